#!/usr/bin/env python3
"""debug: forward of the seeded 20-graph batch (tests/test_gpu_parity.py::test_forward_vs_oracle_seeded_batch) with a
library variant (TSDIFF_LIB) against the oracle; prints where the error is"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, engine, synth
if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["TSDIFF_LIB"])
from oracle import tsdiff_oracle as O
from tests.test_gpu_parity import make_model, run_forward, to_dev
for form in os.environ.get("FORMS", "default").split(","):
    engine.OPTIONS.one_launch = form != "perblock"
    engine.OPTIONS.fused_encoder = "force" if form == "fused" else True
    dev = torch.device("cuda:0")
    cfg = synth.DEFAULT_MODEL_CONFIG
    b = synth.wb97xd3_like_batch(20, seed=5)
    b["pos"] = (b["pos"] * np.repeat(np.linspace(0.7, 9.0, 20).astype(np.float32), b["num_nodes_per_graph"])[:, None])
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    sd = O.to_torch_state(synth.synth_state_dict(cfg, 3))
    o_inv, o_ei, o_el = O.forward(sd, cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"],
                                  t["bond_type"], b["num_nodes_per_graph"])
    g = to_dev({**t, "num_graphs": 20}, dev)
    model = make_model(cfg, 3, dev)
    inv, ei, el = run_forward(model, g, dev)
    d = (inv.cpu() - o_inv).abs().view(-1)
    mx = float(o_inv.abs().max())
    k = torch.argsort(d, descending=True)[:12]
    gidx = torch.from_numpy(b["batch"])[o_ei[0][k]]
    print(form, "max|ref|", mx, "rel err", float(d.max()) / mx)
    for i in k.tolist():
        print("  edge", i, "graph", int(torch.from_numpy(b["batch"])[o_ei[0][i]]), "len", float(o_el[i]), "gpu", float(inv[i]), "ref", float(o_inv[i]), "diff", float(d[i]))
    per_graph = torch.zeros(20).scatter_reduce(0, torch.from_numpy(b["batch"])[o_ei[0]], d, "amax")
    print("  per-graph max diff:", [f"{x:.1e}" for x in per_graph.tolist()])
    inv2, _, _ = run_forward(model, g, dev)
    print("  deterministic:", bool(torch.equal(inv, inv2)))
