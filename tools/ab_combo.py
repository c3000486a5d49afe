#!/usr/bin/env python3
"""A/B timing of the per-block launch (layer_combo_kernel) for kernel variants built as separate libraries.

    TSDIFF_LIB=/path/to/variant.so python tools/ab_combo.py [c2|c5|both]

Prints the average duration (us) of: a filter-only launch, a node-only launch, the combined launch, and the mean over
the L+1 launches of a forward -- HIP events on the launch stream, interleaved rounds (min and median)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, synth  # noqa: E402

if os.environ.get("TSDIFF_LIB"):
    _lib.LIB_PATH = os.environ["TSDIFF_LIB"]
from bench import make_models, to_dev  # noqa: E402
from tsdiff_amd.sampler import EnsembleSampler  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    dev = torch.device("cuda:0")
    lib = _lib.load()
    cfg = synth.DEFAULT_MODEL_CONFIG
    H, L = 256, 7
    model = make_models(cfg, [0], dev)[0]
    out = {}
    for name in (["c2", "c5"] if which == "both" else [which]):
        if name == "c2":
            g = to_dev(synth.wb97xd3_like_batch(100, seed=1000), dev)
            G = 100
            g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
            reps = 30
        else:
            g = to_dev(synth.dense_stress_batch(1024, n=64, seed=1000), dev)
            G = 1024
            reps = 3
        s = EnsembleSampler([model])
        with torch.no_grad():
            s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
        db = s._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
        N, PU = db.N, db.P // 2
        ea = torch.randn(max(PU, 1), H, device=dev)
        wf = torch.randn(2, max(PU, 1), H, device=dev)
        xa, xb = torch.randn(N, H, device=dev), torch.empty(N, H, device=dev)
        hbuf = torch.randn(N, H, device=dev)

        def blk(layer, fl, xi, xo):
            _lib.check(lib.tsd_interaction_block(
                C.byref(db.cfg), _lib.ptr(db.weights[0]), layer, N, db.enc.struct(),
                _lib.ptr(wf[layer % 2]) if layer >= 0 else None, _lib.ptr(xi), _lib.ptr(hbuf), _lib.ptr(xo), fl, PU,
                db.enc_u.struct(), _lib.ptr(ea), _lib.ptr(wf[fl % 2]) if fl >= 0 else None, _lib.stream_ptr()))

        def forward_blocks():
            blk(-2, 0, xa, xb)
            for l in range(L):
                blk(l, l + 1 if l + 1 < L else -1, xa if l % 2 == 0 else xb, xb if l % 2 == 0 else xa)
        cases = {"filter_only": lambda: blk(-2, 1, xa, xb), "node_only": lambda: blk(2, -1, xa, xb),
                 "combined": lambda: blk(2, 3, xa, xb), "forward/8": forward_blocks}
        res = {k: [] for k in cases}
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for k, fn in cases.items():
            fn()
        torch.cuda.synchronize()
        for rnd in range(5):
            for k, fn in cases.items():
                ev0.record()
                for _ in range(reps):
                    fn()
                ev1.record()
                torch.cuda.synchronize()
                res[k].append(ev0.elapsed_time(ev1) / reps * 1e3 / (8 if k == "forward/8" else 1))
        Eu, E = db.enc_u.num_edges(), db.enc.num_edges()
        flops = (L * (Eu * (4.0 * H * H + H) + E * 2.0 * H + N * 6.0 * H * H)) / (L + 1)
        line = f"{name}: " + "  ".join(f"{k} {min(v):.1f}/{float(np.median(v)):.1f}" for k, v in res.items())
        line += f"   [us min/median]  frac(forward/8, median) = {flops / (float(np.median(res['forward/8'])) * 1e-6) / 157.3e12:.3f}"
        print(line)
        del ea, wf, xa, xb, hbuf, db, s
        model._batches.clear()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
