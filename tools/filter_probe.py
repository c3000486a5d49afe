#!/usr/bin/env python3
"""filter-only launches (tsd_interaction_block16, layer -2) at a given pair count: time per launch, per 64 pairs.
   TSDIFF_LIB=... python tools/filter_probe.py [graphs]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, synth
if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["TSDIFF_LIB"])
from bench import make_models, to_dev
from tsdiff_amd.sampler import EnsembleSampler
G = int(sys.argv[1]) if len(sys.argv) > 1 else 800
dev = torch.device("cuda:0")
lib = _lib.load()
model = make_models(synth.DEFAULT_MODEL_CONFIG, [0], dev)[0]
g = to_dev(synth.wb97xd3_like_batch(G, seed=1000), dev)
g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
s = EnsembleSampler([model])
with torch.no_grad():
    s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
db = s._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
N, PU, H = db.N, db.P // 2, 256
Eu = db.enc_u.num_edges()
ea = torch.randn(max(PU, 1), H, device=dev)
ea16 = torch.empty_like(ea)
_lib.check(lib.tsd_attr_planes(H, ea.shape[0], _lib.ptr(ea), _lib.ptr(ea16), None, _lib.stream_ptr()))
wf = torch.empty(max(PU, 1), H, device=dev)
x = torch.randn(N, H, device=dev)
def launch():
    _lib.check(lib.tsd_interaction_block16(C.byref(db.cfg), _lib.ptr(db.weights16[0]), -2, N, db.enc.struct(), None, _lib.ptr(x), _lib.ptr(x),
                                           _lib.ptr(x), 3, PU, db.enc_u.struct(), _lib.ptr(ea16), _lib.ptr(wf), None, _lib.stream_ptr()))
for _ in range(20): launch()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    ev0.record()
    for _ in range(50): launch()
    ev1.record(); torch.cuda.synchronize()
    best = min(best, ev0.elapsed_time(ev1) / 50 * 1e3)
print(f"{os.environ.get('TSDIFF_LIB', 'default')}: graphs {G}, undirected edges {Eu}: filter-only launch {best:.1f} us = {best * 512 / (Eu / 64):.2f} slot-us per 64 pairs "
      f"(MFMA-ideal {Eu * 2 * 2 * 256 * 256 * 3 / 2.5e15 * 1e6:.1f} us per launch)")
