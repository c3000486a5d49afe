#!/usr/bin/env python3
"""Split-f16 forward (TSDIFF_GEMM=h2, csrc/split16.hpp) against the fp32-MFMA forward and the CPU oracle in fp32 and
fp64: the same batches as tests/tools/parity_report.py, both arithmetic modes of the same library in one process.
    python tests/tools/check_h2.py [n_seeds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdiff_oracle as O  # noqa: E402  (checker only)
from tsdiff_amd import engine, synth  # noqa: E402
from tsdiff_amd.epsnet import get_model  # noqa: E402
from tsdiff_amd.sampler import EnsembleSampler  # noqa: E402
from tsdiff_amd.utils import AttrDict  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.DEFAULT_MODEL_CONFIG
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print("# edge_inv of the split-f16 forward vs the fp32-MFMA forward vs the CPU oracle (full model, 24-graph batches)\n")
print("| seed | edges | h2 - fp64 | f32 - fp64 | oracle fp32 - fp64 | h2 - f32 | h2 - oracle fp32 |")
print("|---|---:|---:|---:|---:|---:|---:|")
for seed in range(nseeds):
    sd_np = synth.synth_state_dict(cfg, seed)
    model = get_model(AttrDict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    model = model.to(dev)
    b = synth.wb97xd3_like_batch(24, seed=100 + seed)
    scale = np.repeat(np.random.default_rng(seed).uniform(0.6, 9.0, 24).astype(np.float32), b["num_nodes_per_graph"])
    b["pos"] = (b["pos"] * scale[:, None]).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    o32 = O.forward(O.to_torch_state(sd_np), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"],
                    t["bond_type"], b["num_nodes_per_graph"])[0].view(-1).numpy().astype(np.float64)
    o64 = O.forward(O.to_torch_state(sd_np, torch.float64), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].double(),
                    t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])[0].view(-1).numpy()
    g = {k: v.to(dev) for k, v in t.items()}
    res = {}
    for mode in ("h2", "f32"):
        engine.OPTIONS.gemm = mode
        with torch.no_grad():
            inv, ei, el = model(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                g["batch"], torch.zeros(24, dtype=torch.long, device=dev))
        res[mode] = inv.view(-1).cpu().numpy().astype(np.float64)
        db = model._batches[0][2]
        assert db.gemm_mode() == mode, (db.gemm_mode(), mode)
    sc = np.abs(o64).max()
    e = lambda a, r: np.abs(a - r).max() / sc
    print(f"| {seed} | {len(o64)} | {e(res['h2'], o64):.2e} | {e(res['f32'], o64):.2e} | {e(o32, o64):.2e} | "
          f"{e(res['h2'], res['f32']):.2e} | {e(res['h2'], o32):.2e} |")

# ---- 20 LD steps at batch 100 in both modes (same injected noise): trajectories side by side ----
models = []
for s in range(1):
    m = get_model(AttrDict(cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, s).items()}, strict=False)
    models.append(m.to(dev))
b = synth.wb97xd3_like_batch(100, seed=1000)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items() if isinstance(v, np.ndarray)}
N = g["pos"].shape[0]
gen = torch.Generator(device=dev)
gen.manual_seed(7)
pos0 = torch.randn(N, 3, device=dev, generator=gen) * 1.5
noises = torch.randn(20, N, 3, device=dev, generator=gen)
out = {}
for mode in ("h2", "f32"):
    engine.OPTIONS.gemm = mode
    sampler = EnsembleSampler(models)
    pos, traj = sampler.dynamic_sampling(g["atom_type"], g["r_feat"], g["p_feat"], pos0, g["bond_index"], g["bond_type"],
                                         g["batch"], 100, extend_order=True, n_steps=20, step_lr=1e-7, clip=1000,
                                         sampling_type="ld", noises=noises)
    out[mode] = pos.double().cpu().numpy()
d = np.abs(out["h2"] - out["f32"]).max() / np.abs(out["f32"]).max()
print(f"\n20 LD steps at batch 100 (injected noise): max |pos_h2 - pos_f32| / max |pos| = {d:.2e}")
