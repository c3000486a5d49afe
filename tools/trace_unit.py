#!/usr/bin/env python3
"""Where a workgroup of the fused per-unit encoder (csrc/kernels_unit.hip) spends its time, from a -DTSD_UNIT_TRACE
variant build (shader-clock cycles of wave 0 per phase, summed over the tiles and blocks of the unit):
    tools/build_variant.sh utrace "-DTSD_UNIT_TRACE" kernels_unit.hip && python tools/trace_unit.py [c5|ens8|gNNN]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, engine, synth
_lib.LIB_PATH = os.environ.get("TSDIFF_LIB", os.path.join(ROOT, "tools", "bin", "lib_utrace.so"))
from bench import make_models, to_dev
from tsdiff_amd.sampler import EnsembleSampler
workload = sys.argv[1] if len(sys.argv) > 1 else "c5"
engine.OPTIONS.fused_encoder = "force"
dev = torch.device("cuda:0")
lib = _lib.load()
dbg = C.CDLL(_lib.LIB_PATH).tsd_debug_unit_trace
dbg.argtypes = [C.c_void_p]
cfg = synth.DEFAULT_MODEL_CONFIG
M = 8 if workload == "ens8" else 1
models = make_models(cfg, range(M), dev)
if workload == "c5":
    G = 256  # one unit per CU: the numbers of a resident workgroup
    b = synth.dense_stress_batch(G, n=64, seed=1000)
else:
    G = int(workload[1:]) if workload[0] == "g" else 100
    b = synth.wb97xd3_like_batch(G, seed=1000)
g = to_dev(b, dev)
if workload != "c5":
    g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
s = EnsembleSampler(models)
def fwd():
    with torch.no_grad():
        s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
for _ in range(3): fwd()
torch.cuda.synchronize()
buf = np.zeros(4096 * 16, dtype=np.uint64)
fwd(); torch.cuda.synchronize()
assert dbg(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64).reshape(-1, 16)
t = t[t[:, 9] > 0]
pp = os.environ.get("TSDIFF_PINGPONG", "0") != "0" and workload == "c5"
if pp:
    # ping-pong form: rows alternate team 0 / team 1; slots: 0 GEMM nn.0, 1 ssp, 2 GEMM nn.2, 3 filter tile -> LDS, 4 accumulate
    # (+ fetch), 5 next planes, 6 node chain, 7 idle / waiting for the slot (team skew, barriers of the other team)
    nm = ["GEMM nn.0 (3 parts)", "ssp epilogue (3 parts)", "GEMM nn.2 (3 parts)", "filter tile -> LDS", "accumulate + fetch",
          "next planes + ring start", "node chain", "between cycles"]
    for team in (0, 1):
        tt = t[team::2]
        tl = np.maximum(tt[:, 8], 1)
        print(f"team {team}: {len(tt)} workgroups, total cycles median {np.median(tt[:, 9]):.0f} = {np.median(tt[:, 9]) / clk if False else np.median(tt[:, 9]) / 2400:.1f} us at 2.4 GHz, tiles x blocks {np.median(tt[:, 8]):.0f}")
        for i, n in enumerate(nm):
            per = tt[:, i] / (tl if i != 6 else cfg["encoder"]["num_convs"])
            print(f"  {n:28s} {np.median(per):9.0f} cycles per {'tile' if i != 6 else 'block'}   share {np.median(tt[:, i] / tt[:, 9]):6.1%}")
    sys.exit(0)
names = ["convert attr -> planes", "GEMM nn.0", "ssp epilogue", "GEMM nn.2 + prefetch", "filter tile -> LDS", "accumulate", "node chain", "staging"]
clk = 2.4e3  # cycles per us at 2.4 GHz (nominal; the effective clock is lower under load)
print(f"{len(t)} workgroups; total cycles per workgroup median {np.median(t[:, 9]):.0f} = {np.median(t[:, 9]) / clk:.1f} us at 2.4 GHz; "
      f"tile x block count median {np.median(t[:, 8]):.0f}")
tiles = np.maximum(t[:, 8], 1)
for i, n in enumerate(names):
    per = t[:, i] / (tiles if i < 6 else cfg["encoder"]["num_convs"] if i == 6 else 1)
    unit = "per tile" if i < 6 else ("per block" if i == 6 else "per unit")
    print(f"  {n:26s} {np.median(per):9.0f} cycles {unit} = {np.median(per) / clk:6.2f} us   share {np.median(t[:, i] / t[:, 9]):6.1%}")
