#!/usr/bin/env python3
"""Accuracy of the split-f16 forward (csrc/split16.hpp) against an fp64 evaluation of the pinned oracle, next to the
fp32-input-MFMA forward, over weight scale, weight distribution, hidden size and geometry scale -- the table of
profiles/r04_split_f16_sweep.md:

    python tests/tools/split_f16_sweep.py [--graphs 100] > profiles/r04_split_f16_sweep.md

Per case: e = max|x - ref64| / max|ref64| of edge_inv for both arithmetic forms; the same on the 1 % of entries with
the smallest |ref64| (absolute error over the tensor's scale: what a small entry loses); `per-element` = max |x - ref64| / |ref64| over the entries with |ref64| >= 1 % of the
tensor's scale; and whether the split-f16 call left the f16 range and was rerun in fp32 (TSD_STATUS_RANGE -> `fallback`).
tests/test_gpu_round4.py runs a reduced sweep with the assertions."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def scaled_state_dict(cfg, seed, scale=1.0, heavy_tail=False, names_only=None):
    """synth weights with every dense weight matrix (ndim == 2 Linear weights, not the embeddings) times `scale`;
    heavy_tail: Student-t (3 degrees of freedom) draws of the same standard deviation per tensor instead of uniform ones"""
    from tsdiff_amd import synth
    sd = {k: v.copy() for k, v in synth.synth_state_dict(cfg, seed).items()}
    rng = np.random.default_rng(seed + 77)
    for k, v in sd.items():
        dense = v.ndim == 2 and "emb" not in k
        if heavy_tail and dense:
            t = rng.standard_t(3, size=v.shape).astype(np.float32)
            sd[k] = (t * (v.std() / max(t.std(), 1e-12))).astype(np.float32)
        if dense and (names_only is None or any(n in k for n in names_only)):
            sd[k] = (sd[k] * np.float32(scale)).astype(np.float32)
    return sd


def trained_state_dict(cfg, steps, dev, graphs=200, seed=0, lr=5e-4):
    """weights the package trained ITSELF: `steps` optimizer steps of its default training step (split-f16 fused step, Adam
    of configs/train_config.yml, clip 3000: reference train.py:124-152) from the closed-form initialisation over 8 rotating
    synthetic batches with ground-truth geometries (bond-length 1.5 x the generator's coordinates, as bench.py's training
    workload).  -> (state dict as numpy arrays incl. the schedule buffers, range trips, [mean loss per step])"""
    from types import SimpleNamespace
    from tsdiff_amd import optim, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    model = get_model(AttrDict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, seed).items()}, strict=False)
    model = model.to(dev)
    model.train()
    batches = []
    for k in range(8):
        b = synth.wb97xd3_like_batch(graphs, seed=4000 + k)
        g = {kk: torch.from_numpy(v).to(dev) for kk, v in b.items() if isinstance(v, np.ndarray)}
        g["pos"] = (g["pos"] * 1.5).contiguous()
        batches.append(g)
    opt = optim.get_optimizer(SimpleNamespace(type="adam", lr=lr, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
    gen = torch.Generator(device="cpu").manual_seed(4321)
    losses = []
    for it in range(steps):
        g = batches[it % 8]
        ts = torch.randint(0, 5000, (graphs,), generator=gen).to(dev)
        noise = torch.randn(g["pos"].shape, generator=gen).to(dev)
        opt.zero_grad()
        loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
                              g["num_nodes_per_graph"], graphs, _time_step=ts, _pos_noise=noise)
        m = loss.mean()
        m.backward()
        optim.clip_grad_norm_(model.parameters(), 3000.0)
        opt.step()
        losses.append(float(m.detach()))
    model.eval()
    sd = {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}
    return sd, int(getattr(model, "_h2_range_trips", 0)), losses


def rel_elementwise(x, ref, floor_frac=0.01):
    """largest |x - ref| / |ref| over the entries with |ref| above `floor_frac` of the tensor's scale (the per-element
    relative error where it means something), and the number of such entries"""
    x, ref = np.asarray(x, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    big = np.abs(ref) >= floor_frac * np.abs(ref).max()
    return float((np.abs(x - ref)[big] / np.abs(ref)[big]).max()), int(big.sum())


def run_case(cfg, sd_np, graphs, seed, pos_lo, pos_hi, dev):
    """-> dict(e_f32, e_h2, small_f32, small_h2, fallback, scale, preflight)"""
    from oracle import tsdiff_oracle as O  # the checker
    from tsdiff_amd import engine, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    b = synth.wb97xd3_like_batch(graphs, seed=seed)
    scale = np.repeat(np.linspace(pos_lo, pos_hi, graphs).astype(np.float32), b["num_nodes_per_graph"])[:, None]
    b["pos"] = (b["pos"] * scale).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    o64 = O.forward(O.to_torch_state(sd_np, torch.float64), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].double(),
                    t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])[0].numpy().reshape(-1)
    g = {k: v.to(dev) for k, v in t.items()}
    out = {}
    old = engine.OPTIONS.gemm
    try:
        for mode in ("f32", "h2"):
            engine.OPTIONS.gemm = mode
            model = get_model(AttrDict(cfg))
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
            model = model.to(dev)
            with torch.no_grad():
                inv, _, _ = model(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
                                  torch.zeros(graphs, dtype=torch.long, device=dev))
            db = model._batches[0][2]
            out[mode] = inv.cpu().numpy().reshape(-1).astype(np.float64)
            if mode == "h2":
                out["fallback"] = db.gemm == "f32"
                out["preflight"] = model.preflight_split_f16()
    finally:
        engine.OPTIONS.gemm = old
    sc = max(float(np.abs(o64).max()), 1e-300)
    small = np.argsort(np.abs(o64))[: max(1, o64.size // 100)]
    res = {"scale": sc, "fallback": out["fallback"], "preflight": out["preflight"], "finite": bool(np.isfinite(o64).all())}
    for mode in ("f32", "h2"):
        d = np.abs(out[mode] - o64)
        res["e_" + mode] = float(d.max() / sc)
        res["small_" + mode] = float(d[small].max() / sc)
        res["elem_" + mode] = rel_elementwise(out[mode], o64)[0]
    return res


CASES = [  # name, hidden, convs, weight scale, heavy tail, geometry scale range (x the generator's ~1.5 A coordinates)
    ("default", 256, 7, 1.0, False, (0.7, 9.0)),
    ("weights x 0.1", 256, 7, 0.1, False, (0.7, 9.0)),
    ("weights x 0.5", 256, 7, 0.5, False, (0.7, 9.0)),
    ("weights x 2", 256, 7, 2.0, False, (0.7, 9.0)),
    ("weights x 4", 256, 7, 4.0, False, (0.7, 9.0)),
    ("heavy-tailed weights (Student t, 3 dof)", 256, 7, 1.0, True, (0.7, 9.0)),
    ("compact geometries 0.3 - 1", 256, 7, 1.0, False, (0.3, 1.0)),
    ("stretched geometries 4 - 12", 256, 7, 1.0, False, (4.0, 12.0)),
    ("hidden 128", 128, 4, 1.0, False, (0.7, 9.0)),
    ("hidden 64", 64, 3, 1.0, False, (0.7, 9.0)),
    ("hidden 64, weights x 0.1", 64, 3, 0.1, False, (0.7, 9.0)),
]


def config_for(hidden, convs):
    from tsdiff_amd import synth
    return synth.DEFAULT_MODEL_CONFIG if (hidden, convs) == (256, 7) else synth.small_model_config(hidden, convs)


def main():
    graphs = int(sys.argv[sys.argv.index("--graphs") + 1]) if "--graphs" in sys.argv else 100
    dev = torch.device("cuda:0")
    print("# Split-f16 forward against an fp64 evaluation, next to the fp32-input-MFMA forward (tests/tools/split_f16_sweep.py)\n")
    print(f"`edge_inv` of a {graphs}-graph wb97xd3-like batch; e = max|x - ref64| / max|ref64|; `small 1 %` = the same maximum over "
          "the 1 % of entries with the smallest |ref64| (absolute error over the tensor scale); `fallback` = the split-f16 call "
          "left the f16 range (TSD_STATUS_RANGE) and was rerun on the fp32-MFMA kernels, so both columns are that path; "
          "`< 2^-14` = packed weights (folded matrices included) below the f16 normal range (tsd_weights16_preflight).\n")
    print("| case | max abs ref64 | e fp32 MFMA | e split-f16 | small 1 % fp32 | small 1 % split-f16 | per-element fp32 | per-element split-f16 | max abs w | w < 2^-14 | fallback |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    cases = [(n, H, L, scaled_state_dict(config_for(H, L), 3, ws, heavy), rng_) for n, H, L, ws, heavy, rng_ in CASES]
    # weights the package trained itself (round 6): 300 and 1000 optimizer steps of its own default training step
    for steps in (300, 1000):
        sd, trips, losses = trained_state_dict(config_for(256, 7), steps, dev)
        cases.append((f"trained by the package's own step, {steps} steps (loss {losses[0]:.0f} -> {np.mean(losses[-8:]):.0f}, "
                      f"{trips} range trips)", 256, 7, sd, (0.7, 9.0)))
    for name, H, L, sd, (lo, hi) in cases:
        cfg = config_for(H, L)
        r = run_case(cfg, sd, graphs, 1000, lo, hi, dev)
        pf = r["preflight"]
        print(f"| {name} | {r['scale']:.3e} | {r['e_f32']:.2e} | {r['e_h2']:.2e} | {r['small_f32']:.2e} | {r['small_h2']:.2e} | "
              f"{r['elem_f32']:.2e} | {r['elem_h2']:.2e} | {pf['max_abs']:.3g} | {pf['below_f16_normal']} of {pf['count']} | {'yes' if r['fallback'] else 'no'} |", flush=True)


if __name__ == "__main__":
    main()
