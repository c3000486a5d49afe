#!/usr/bin/env python3
"""debug: one-launch vs per-block forward of the seeded 20-graph batch, bitwise, for a library variant (TSDIFF_LIB)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, engine, synth
if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["TSDIFF_LIB"])
from tests.test_gpu_parity import make_model, run_forward, to_dev
dev = torch.device("cuda:0")
cfg = synth.DEFAULT_MODEL_CONFIG
b = synth.wb97xd3_like_batch(20, seed=5)
b["pos"] = (b["pos"] * np.repeat(np.linspace(0.7, 9.0, 20).astype(np.float32), b["num_nodes_per_graph"])[:, None])
t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
g = to_dev({**t, "num_graphs": 20}, dev)
res = {}
for form in ("mega", "perblock"):
    engine.OPTIONS.one_launch = form == "mega"
    model = make_model(cfg, 3, dev)
    inv, ei, el = run_forward(model, g, dev)
    res[form] = inv.clone()
d = (res["mega"] - res["perblock"]).abs().view(-1)
print(os.environ.get("TSDIFF_LIB", "head"), "mega == perblock:", bool(torch.equal(res["mega"], res["perblock"])), "max diff", float(d.max()),
      "n diff", int((d > 0).sum()), "of", d.numel())
from tests.test_gpu_round4 import _db
res_u = {}
for form in ("mega", "perblock"):
    engine.OPTIONS.one_launch = form == "mega"
    model = make_model(cfg, 3, dev)
    inv, ei, el = run_forward(model, g, dev)
    db = _db(model)
    n = db.out_u.num_edges()
    res_u[form] = db.edge_inv_u.view(-1)[:n].clone()
    print(form, "out_u edges", n, "node tiles", (db.N + 15) // 16, "N", db.N)
du = (res_u["mega"] - res_u["perblock"]).abs()
idx = torch.nonzero(du > 0).view(-1).tolist()
print("differing undirected indices:", idx)
print("tile/row:", [(i // 32, i % 32) for i in idx])
PU = db.P // 2
ar = db.attr_row[:n].cpu()
src = db.out_u.src[:n].cpu(); dst = db.out_u.dst[:n].cpu(); dist = db.out_u.dist[:n].cpu()
print("PU", PU, "diff_u edges", db.diff_u.num_edges(), "enc_u", db.enc_u.num_edges())
for i in idx:
    print(" u", i, "src", int(src[i]), "dst", int(dst[i]), "attr_row", int(ar[i]), ">=PU" if int(ar[i]) >= PU else "", "dist", float(dist[i]),
          "mega", float(res_u["mega"][i]), "perblock", float(res_u["perblock"][i]))
print("rows with attr_row >= PU:", torch.nonzero(ar >= PU).view(-1).tolist())
