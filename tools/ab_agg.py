#!/usr/bin/env python3
"""A/B timing of the stand-alone CFConv aggregation (cfconv_aggregate_kernel, the HBM-bound form) at configs[4]
size for library variants:   python tools/ab_agg.py NAME=LIB ...   (LIB = path of a variant .so or `default`).
Every variant in its own child process, interleaved over 3 rounds; prints us per launch (min / median) and the
fraction of 8 TB/s for 1028 E + 2048 N + 4 bytes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from tsdiff_amd import _lib, synth
    if lib != "default":
        _lib.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
    from bench import make_models, to_dev
    from tsdiff_amd.sampler import EnsembleSampler
    L = _lib.load()
    dev = torch.device("cuda:0")
    model = make_models(synth.DEFAULT_MODEL_CONFIG, [0], dev)[0]
    g = to_dev(synth.dense_stress_batch(1024, n=64, seed=1000), dev)
    s = EnsembleSampler([model])
    db = s._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
    db.geometry(g["pos"])
    H, N, E = 256, db.N, db.enc.num_edges()
    Wd = torch.randn(E, H, device=dev)
    x1 = torch.randn(N, H, device=dev)
    out = torch.empty(N, H, device=dev)

    def launch():
        _lib.check(L.tsd_cfconv_aggregate(H, N, _lib.ptr(db.enc.row_ptr), _lib.ptr(db.enc.dst), None, _lib.ptr(Wd),
                                          _lib.ptr(x1), _lib.ptr(out), _lib.stream_ptr()))
    for _ in range(10):
        launch()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        ev0.record()
        for _ in range(20):
            launch()
        ev1.record()
        torch.cuda.synchronize()
        ts.append(ev0.elapsed_time(ev1) / 20 * 1e3)
    # single launches (no back-to-back overlap of kernel tails), as rocprof sees them
    singles = []
    for _ in range(10):
        ev0.record()
        launch()
        ev1.record()
        torch.cuda.synchronize()
        singles.append(ev0.elapsed_time(ev1) * 1e3)
    print(f"RESULT {min(ts):.1f} {float(np.median(ts)):.1f} {float(np.median(singles)):.1f} {float(out.double().abs().sum()):.3f}")


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2])
    cfgs = [a.split("=", 1) for a in sys.argv[1:]]
    res = {n: [] for n, _ in cfgs}
    for _ in range(3):
        for n, lib in cfgs:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], capture_output=True, text=True)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(n, "FAILED", o.stderr[-1500:])
                continue
            res[n].append([float(x) for x in line[0].split()[1:]])
    bytes_ = 1028.0 * 4128768 + 2048.0 * 65536 + 4
    for n, v in res.items():
        if v:
            mn = min(x[0] for x in v)
            med = sorted(x[1] for x in v)[len(v) // 2]
            sg = sorted(x[2] for x in v)[len(v) // 2]
            print(f"  {n:16s} back-to-back us min {mn:.1f} median {med:.1f} (frac {bytes_ / (med * 1e-6) / 8e12:.3f})   "
                  f"single launch median {sg:.1f} (frac {bytes_ / (sg * 1e-6) / 8e12:.3f})   checksum {v[0][3]}")


if __name__ == "__main__":
    main()
