#!/usr/bin/env python3
"""Training-drift study of the default (split-f16) arithmetic of the fused training step (review item of round 4).

Three trainings from ONE initialisation over the SAME batches, time steps and noise draws (reference loop: get_loss,
backward, clip_grad_norm_, Adam -- train.py:124-152):

    h2    the default: tile GEMMs on split-f16 operands (OPTIONS.train_gemm = "h2")
    f32   the fp32-input MFMA kernels (the reference arithmetic class)
    ops   fp32 again, but the op-by-op autograd form (OPTIONS.train = "ops"): the same function evaluated by other
          kernels in another summation order -- the NOISE FLOOR: how far two fp32 trainings drift apart by
          re-association alone

and reports, every `--every` optimizer steps, the loss of each and the parameter distances
    || theta_h2 - theta_f32 || / || theta_f32 ||     against     || theta_ops - theta_f32 || / || theta_f32 ||.

    python tools/train_drift.py [--steps 300] [--graphs 200] [--every 50] [--out gpurun_out/r05_train_drift.md]

tests/test_gpu_round5.py::test_train_drift_h2_within_fp32_noise runs `drift()` at 50 steps (smaller batch)."""
import argparse
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def train_run(mode, steps, graphs, every, dev, cfg=None, lr=5e-4, n_batches=8, seed=2000):
    """one training; returns (losses [steps], snapshots {step: flat parameter copy}, range trips)"""
    import bench
    from tsdiff_amd import optim, synth
    from tsdiff_amd.options import OPTIONS
    cfg = cfg or synth.DEFAULT_MODEL_CONFIG
    old = (OPTIONS.train, OPTIONS.train_gemm)
    pert = int(mode[4:]) if mode.startswith("f32p") else 0  # f32p<k>: the fp32 step from a 1-ulp-perturbed initialisation
    OPTIONS.train, OPTIONS.train_gemm = ("ops", "f32") if mode == "ops" else ("fused", "f32" if pert else mode)
    try:
        model = bench.make_models(cfg, [0], dev)[0]
        if pert:
            gp = torch.Generator(device="cpu").manual_seed(900 + pert)
            with torch.no_grad():
                for _, p in sorted(model.named_parameters()):
                    if p.requires_grad:
                        p.mul_((1.0 + 1e-7 * torch.randn(p.shape, generator=gp)).to(p.device))
        model.train()
        batches = []
        for k in range(n_batches):
            g = bench.to_dev(synth.wb97xd3_like_batch(graphs, seed=seed + k), dev)
            g["pos"] = (g["pos"] * 1.5).contiguous()
            batches.append(g)
        opt = optim.get_optimizer(SimpleNamespace(type="adam", lr=lr, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
        gen = torch.Generator(device="cpu").manual_seed(1234)
        losses, snaps = [], {}
        names = [n for n, p in model.named_parameters() if p.requires_grad]

        def flat():
            P = dict(model.named_parameters())
            return torch.cat([P[n].detach().reshape(-1).double() for n in names])
        snaps[0] = flat()
        for it in range(steps):
            g = batches[it % n_batches]
            ts = torch.randint(0, 5000, (graphs,), generator=gen).to(dev)
            noise = torch.randn(g["pos"].shape, generator=gen).to(dev)
            opt.zero_grad()
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                  g["batch"], g["num_nodes_per_graph"], graphs, _time_step=ts, _pos_noise=noise)
            m = loss.mean()
            m.backward()
            optim.clip_grad_norm_(model.parameters(), 3000.0)
            opt.step()
            losses.append(float(m.detach()))  # (after backward: a range trip rewrites the loss tensor in place)
            if (it + 1) % every == 0 or it + 1 == steps:
                snaps[it + 1] = flat()
        return losses, snaps, int(getattr(model, "_h2_range_trips", 0))
    finally:
        OPTIONS.train, OPTIONS.train_gemm = old


def drift(steps, graphs, every, dev, cfg=None, perturbed=0):
    """rows [(step, loss_f32, loss_h2, loss_ops, d_h2, d_ops)], trips, relative distance travelled, per-mode results.
    `perturbed` > 0: that many extra fp32 trainings from initialisations perturbed by one ulp (modes f32p1 ..): how far the
    TRAINING DYNAMICS carry a 1e-7 difference, whatever its source -- res[\"f32pK\"] and `perturbed_rows(res)`"""
    res = {m: train_run(m, steps, graphs, every, dev, cfg) for m in ("f32", "h2", "ops") + tuple(f"f32p{k + 1}" for k in range(perturbed))}
    rows = []
    for s in sorted(res["f32"][1]):
        ref = res["f32"][1][s]
        nr = float(ref.norm())
        d = {m: float((res[m][1][s] - ref).norm()) / nr for m in ("h2", "ops")}
        li = max(s - 1, 0)
        rows.append((s, res["f32"][0][li], res["h2"][0][li], res["ops"][0][li], d["h2"], d["ops"]))
    moved = float((res["f32"][1][max(res["f32"][1])] - res["f32"][1][0]).norm()) / float(res["f32"][1][0].norm())
    return rows, res["h2"][2], moved, res


def perturbed_rows(res):
    """{step: [|| theta_f32pK - theta_f32 || / || theta_f32 || for K = 1 ..]}"""
    ks = sorted(m for m in res if m.startswith("f32p"))
    out = {}
    for s in sorted(res["f32"][1]):
        ref = res["f32"][1][s]
        out[s] = [float((res[m][1][s] - ref).norm()) / float(ref.norm()) for m in ks]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--perturbed", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_train_drift.md"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, trips, moved, res = drift(a.steps, a.graphs, a.every, dev, perturbed=a.perturbed)
    pr = perturbed_rows(res)
    L = ["# Training drift of the split-f16 step against fp32 (tools/train_drift.py)", "",
         f"{a.steps} optimizer steps at batch {a.graphs} (Adam lr 5e-4, betas 0.95 / 0.999, clip 3000: configs/train_config.yml), "
         "8 synthetic batches rotating, the SAME batches / time steps / noise draws in every run, one initialisation "
         "(closed-form synthetic weights).  `h2` = the default split-f16 step, `f32` = fp32-input MFMA step, `ops` = fp32 "
         "op-by-op autograd form (other kernels, other summation order: the re-association noise floor).  Distances are "
         "|| theta_x - theta_f32 || / || theta_f32 || over all trainable parameters; the parameters themselves moved "
         f"{moved:.3e} (relative) from the initialisation over the run.  Split-f16 range trips: {trips}.", "",
         f"`f32p1..{a.perturbed}`: the SAME fp32 step from initialisations perturbed by one ulp (theta (1 + 1e-7 xi)): what the "
         "training dynamics make of a 1e-7 difference.", "",
         "| step | loss f32 | loss h2 | loss ops | h2 vs f32 | ops vs f32 (noise floor) | ratio | 1-ulp-perturbed fp32 runs vs f32 | h2 / largest perturbed |",
         "|---:|---:|---:|---:|---:|---:|---:|---|---:|"]
    for s, lf, lh, lo, dh, do in rows:
        pp = pr.get(s, [])
        L.append(f"| {s} | {lf:.6g} | {lh:.6g} | {lo:.6g} | {dh:.3e} | {do:.3e} | {dh / max(do, 1e-300):.2f} | "
                 + " ".join(f"{x:.3e}" for x in pp) + f" | {dh / max(max(pp), 1e-300) if pp and max(pp) > 0 else 0:.2f} |")
    lf, lh, lo = res["f32"][0], res["h2"][0], res["ops"][0]
    import numpy as np
    rel = lambda x, y: float(np.max(np.abs(np.array(x) - np.array(y)) / np.maximum(np.abs(np.array(y)), 1e-30)))
    L += ["", f"Largest relative difference of the per-step mean loss over the run: h2 vs f32 {rel(lh, lf):.3e}, ops vs f32 "
              f"{rel(lo, lf):.3e}."]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    open(a.out, "w").write("\n".join(L) + "\n")
    print("\n".join(L))


if __name__ == "__main__":
    main()
