#!/usr/bin/env python3
"""one-launch forward vs launch-per-block forward (both split-f16): workspace tensors side by side"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, engine, synth
if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["TSDIFF_LIB"])  # (a variant build: tools/build_variant.sh)
from tsdiff_amd.epsnet import get_model
from tsdiff_amd.utils import AttrDict
dev = torch.device("cuda:0")
cfg = synth.DEFAULT_MODEL_CONFIG
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = synth.synth_state_dict(cfg, 0)
model = get_model(AttrDict(cfg)); model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False); model = model.to(dev)
b = synth.wb97xd3_like_batch(G, seed=100)
b["pos"] = (b["pos"] * 1.7).astype(np.float32)
g = {k: torch.from_numpy(v).to(dev) for k, v in b.items() if isinstance(v, np.ndarray)}
H, L = 256, 7
def pad(n): return (n + 63) & ~63
out = {}
for one in (False, True):
    engine.OPTIONS.one_launch = one
    with torch.no_grad():
        inv, ei, el = model(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], torch.zeros(G, dtype=torch.long, device=dev))
    db = model._batches[0][2]
    N, P = db.N, db.P; PU = P // 2
    ws = db.workspace
    o = 0
    ea = ws[o:o + 2 * PU * H].view(2 * PU, H).clone(); o += pad(2 * PU * H)
    slots = L  # (mega shape)
    wf = ws[o:o + slots * PU * H].view(slots, PU, H).clone(); o += pad(slots * PU * H)
    h = ws[o:o + N * H].view(N, H).clone(); o += pad(N * H)
    x1 = ws[o:o + N * H].view(N, H).clone(); o += pad(N * H)
    x1b = ws[o:o + N * H].view(N, H).clone(); o += pad(N * H)
    Eu = db.enc_u.num_edges()
    out[one] = dict(inv=inv.clone(), ea=ea, wf=wf, h=h, x1=x1, x1b=x1b, Eu=Eu, st=int(db.status[0].item()))
a, m = out[False], out[True]
print("status words", a["st"], m["st"], "Eu", a["Eu"], "N", N, "PU", PU)
def cmp(name, x, y):
    d = (x.double() - y.double()).abs()
    print(f"{name:10s} max|d| {float(d.max()):.3e}  scale {float(x.abs().max()):.3e}  rows differing {int((d.max(-1).values > 0).sum()) if d.dim() > 1 else int((d > 0).sum())} / {x.shape[0]}")
cmp("edge_inv", a["inv"].view(-1), m["inv"].view(-1))
Eu = a["Eu"]
cmp("ea[:Eu]", a["ea"][:Eu], m["ea"][:Eu])
cmp("ea[PU:]", a["ea"][PU:], m["ea"][PU:])
cmp("wf0", a["wf"][0][:Eu], m["wf"][0][:Eu])
for l in range(1, L):
    cmp(f"wf{l} (pb slot {l%2})", a["wf"][l % 2][:Eu] if l >= L - 2 else m["wf"][l][:Eu], m["wf"][l][:Eu])
cmp("h", a["h"], m["h"])
cmp("x1(last odd)", a["x1"], m["x1"]); cmp("x1b", a["x1b"], m["x1b"])
