import ctypes as C, os, sys, time
import numpy as np, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, engine, synth
_lib.LIB_PATH = os.environ.get("TSDIFF_LIB", os.path.join(ROOT, "tools", "bin", "lib_utrace.so"))
from bench import make_models, to_dev
from tsdiff_amd.sampler import EnsembleSampler
dev = torch.device("cuda:0")
lib = _lib.load()
dbg = C.CDLL(_lib.LIB_PATH).tsd_debug_unit_trace
dbg.argtypes = [C.c_void_p]
cfg = synth.DEFAULT_MODEL_CONFIG
models = make_models(cfg, [0], dev)
for G in (32, 64, 128, 256, 512, 1024):
    b = synth.dense_stress_batch(G, n=64, seed=1000)
    g = to_dev(b, dev)
    s = EnsembleSampler(models)
    models[0]._batches.clear()
    def fwd():
        with torch.no_grad():
            s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
    for _ in range(2): fwd()
    db = models[0]._batches[0][2]
    bs = db.struct()
    L = 7
    def enc():
        _lib.check(lib.tsd_forward_encoder(C.byref(db.cfg), C.byref(bs), 0, L, _lib.stream_ptr()))
    for _ in range(3): enc()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(4): enc()
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 4
    buf = np.zeros(4096 * 16, dtype=np.uint64)
    assert dbg(buf.ctypes.data_as(C.c_void_p)) == 0
    t = buf.astype(np.int64).reshape(-1, 16); t = t[t[:, 9] > 0][:G]
    cyc = np.median(t[:, 9])
    rounds = max(1, -(-G // 256))
    print(f"G={G:5d} units: encoder {ms:8.3f} ms, per-workgroup cycles (median) {cyc/1e6:7.3f} M, rounds {rounds}, "
          f"implied clock {cyc * rounds / (ms * 1e-3) / 1e9:5.2f} GHz", flush=True)
