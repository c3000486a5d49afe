#!/usr/bin/env python3
"""Timeline of the one-launch forward from a -DTSD_TRACE variant build (wall_clock64 stamps, 100 MHz):
    tools/build_variant.sh mtrace "-DTSD_MEGA_TRACE" kernels_combo.hip && python tools/trace_mega.py"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, synth
_lib.LIB_PATH = os.environ.get("TSDIFF_LIB", os.path.join(ROOT, "tools", "bin", "lib_mtrace.so"))
from bench import make_models, to_dev
from tsdiff_amd.sampler import EnsembleSampler
dev = torch.device("cuda:0")
lib = _lib.load()
dbg = C.CDLL(_lib.LIB_PATH).tsd_debug_mega_trace
dbg.argtypes = [C.c_void_p]
cfg = synth.DEFAULT_MODEL_CONFIG
model = make_models(cfg, [0], dev)[0]
g = to_dev(synth.wb97xd3_like_batch(100, seed=1000), dev)
g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
s = EnsembleSampler([model])
def fwd():
    with torch.no_grad():
        s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
for _ in range(3): fwd()
torch.cuda.synchronize()
buf = np.zeros(8192 * 4, dtype=np.uint64)
assert dbg(buf.ctypes.data_as(C.c_void_p)) == 0   # clear what the warm-up calls wrote
fwd(); torch.cuda.synchronize()
assert dbg(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64).reshape(-1, 4)
t = t[t[:, 1] > 0]
t0 = t[:, 1].min()
us = lambda x: (x - t0) / 100.0
names = {1: "embed", 2: "node", 3: "filter", 4: "pair", 0: "umap/other"}
print(f"{len(t)} workgroups traced; launch span {us(t[:, 2].max()):.1f} us")
for r in (1, 2, 3, 4, 0):
    m = t[:, 0] == r
    if not m.any(): continue
    st, en = us(t[m, 1]), us(t[m, 2]) if r else us(t[m, 1])
    print(f"{names[r]:10s} n={m.sum():5d} start min/med/max {st.min():7.1f} {np.median(st):7.1f} {st.max():7.1f}   end min/med/max {en.min():7.1f} {np.median(en):7.1f} {en.max():7.1f}  dur med {np.median(en - st):6.1f}")
m = t[:, 0] == 3
if m.any():
    for l in range(1, 7):
        ml = m & (t[:, 3] == l)
        if ml.any():
            en = us(t[ml, 2])
            print(f"  filter layer {l}: start med {np.median(us(t[ml, 1])):7.1f}  end med {np.median(en):7.1f}  p90 {np.percentile(en, 90):7.1f}  "
                  f"p99 {np.percentile(en, 99):7.1f}  max {en.max():7.1f}")
# phase stamps of the node workgroups (variant builds with TSD_MEGA_P_MASK): [tile][block][phase], 100 MHz wall clock
try:
    dph = C.CDLL(_lib.LIB_PATH).tsd_debug_mega_phase
    dph.argtypes = [C.c_void_p]
    pb = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert dph(pb.ctypes.data_as(C.c_void_p)) == 0
    ph = pb.astype(np.int64).reshape(256, 8, 8)
    nt = int((ph[:, 0, 1] > 0).sum()) or int((ph[:, 1, 1] > 0).sum())
    names_p = ["block start", "flags seen", "gathered", "lin2 + barrier", "ssp epilogue", "lin + barrier", "h epilogue + lin1", "x1 published"]
    print(f"node workgroups with stamps: {nt}; per block, medians over tiles (us since the launch's first stamp)")
    for l in range(7):
        row = ph[:nt, l, :]
        have = [p for p in range(8) if (row[:, p] > 0).all()]
        if not have:
            continue
        txt = "  ".join(f"{names_p[p]} {np.median(us(row[:, p])):6.1f}" for p in have)
        print(f"  block {l}: {txt}")
        if len(have) >= 2:
            d = "  ".join(f"{names_p[b]}-{names_p[a]}: med {np.median((row[:, b] - row[:, a]) / 100.0):5.2f} max {((row[:, b] - row[:, a]) / 100.0).max():5.2f}"
                          for a, b in zip(have[:-1], have[1:]))
            print(f"           {d}")
except AttributeError:
    pass
# how many workgroups of each role are running at a time (start/end stamps)
span = us(t[:, 2].max())
print("running workgroups over time (node / filter / pair):")
for x in np.arange(5.0, span, 10.0):
    c = [int(((us(t[:, 1]) <= x) & (us(t[:, 2]) > x) & (t[:, 0] == r)).sum()) for r in (2, 3, 4)]
    print(f"  t = {x:5.0f} us: {c[0]:4d} {c[1]:4d} {c[2]:4d}   (sum {sum(c)})")
