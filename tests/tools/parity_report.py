#!/usr/bin/env python3
"""Error statistics of the HIP forward against the CPU oracle (fp32) and against an fp64 evaluation of the
oracle, over seeded random batches (full H=256 / 7-block model).  Run on the GPU box:
    python tests/tools/parity_report.py > profiles/r01_parity_report.md
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tsdiff_oracle as O  # noqa: E402  (checker only)
from tsdiff_amd import synth  # noqa: E402
from tsdiff_amd.epsnet import get_model  # noqa: E402
from tsdiff_amd.utils import AttrDict  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.DEFAULT_MODEL_CONFIG
rows = []
for seed in range(8):
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
    sd64 = O.to_torch_state(sd_np, torch.float64)
    o64 = O.forward(sd64, cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].double(), t["bond_index"],
                    t["bond_type"], b["num_nodes_per_graph"])[0].view(-1).numpy()
    g = {k: v.to(dev) for k, v in t.items()}
    with torch.no_grad():
        inv, ei, el = model(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                            g["batch"], torch.zeros(24, dtype=torch.long, device=dev))
    hip = inv.view(-1).cpu().numpy().astype(np.float64)
    sc = np.abs(o64).max()
    rows.append((seed, len(hip), np.abs(hip - o32).max() / sc, np.abs(hip - o64).max() / sc,
                 np.abs(o32 - o64).max() / sc, np.median(np.abs(hip - o64) / np.maximum(np.abs(o64), 1e-30))))
print("# edge_inv error statistics, HIP forward vs CPU oracle (full model, 24-graph batches, geometries 0.6-9 A scale)\n")
print("| weights seed | edges | max abs(HIP - oracle fp32) / max abs(ref) | max abs(HIP - oracle fp64) / max abs(ref) | "
      "max abs(oracle fp32 - fp64) / max abs(ref) | median rel. error vs fp64 |")
print("|---|---:|---:|---:|---:|---:|")
for r in rows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.2e} | {r[3]:.2e} | {r[4]:.2e} | {r[5]:.2e} |")
print("\nThe HIP path's distance from the fp64 truth is of the same size as the fp32 CPU evaluation's own.")

# ---- training step: every parameter gradient of the fused step against the oracle's autograd (fp32 and fp64) ----
grows = []
for seed in range(3):
    sd_np = synth.synth_state_dict(cfg, seed)
    model = get_model(AttrDict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
    model = model.to(dev).train()
    b = synth.wb97xd3_like_batch(12, seed=300 + seed)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    t["pos"] = t["pos"] * 1.5
    gen = torch.Generator().manual_seed(seed)
    ts, pn = torch.randint(0, 5000, (12,), generator=gen), torch.randn(t["pos"].shape, generator=gen)
    g = {k: v.to(dev) for k, v in t.items()}
    loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
                          g["num_nodes_per_graph"], 12, _time_step=ts.to(dev), _pos_noise=pn.to(dev))
    loss.mean().backward()
    P = dict(model.named_parameters())
    res = {}
    for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        osd = O.to_torch_state(sd_np, dt) if dt == torch.float64 else O.to_torch_state(sd_np)
        for v in osd.values():
            v.requires_grad_(True)
        ol = O.get_loss(osd, cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].to(dt), t["bond_index"], t["bond_type"],
                        t["batch"], b["num_nodes_per_graph"], ts, pn.to(dt))
        ol.mean().backward()
        res[name] = {k: v.grad.double().numpy() for k, v in osd.items() if v.grad is not None and k not in ("betas", "alphas")}
    worst32 = worst64 = worst_o = 0.0
    wk = ""
    for k, g64 in res["fp64"].items():
        sc = np.abs(g64).max() + 1e-300
        h = P[k].grad.double().cpu().numpy()
        e64 = np.abs(h - g64).max() / sc
        if e64 > worst64:
            worst64, wk = e64, k
        worst32 = max(worst32, np.abs(h - res["fp32"][k]).max() / sc)
        worst_o = max(worst_o, np.abs(res["fp32"][k] - g64).max() / sc)
    grows.append((seed, len(res["fp64"]), worst32, worst64, worst_o, wk))
print("\n# parameter-gradient error statistics, fused training step vs the oracle's autograd (12-graph batches)\n")
print("| weights seed | tensors | worst tensor: max abs(HIP - oracle fp32) / max abs(ref) | worst: HIP vs oracle fp64 | "
      "worst: oracle fp32 vs fp64 | tensor of the worst HIP-vs-fp64 error |")
print("|---|---:|---:|---:|---:|---|")
for r in grows:
    print(f"| {r[0]} | {r[1]} | {r[2]:.2e} | {r[3]:.2e} | {r[4]:.2e} | `{r[5]}` |")
