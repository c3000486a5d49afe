#!/usr/bin/env python3
"""Phase timeline of ONE per-block launch (layer_combo_kernel) from a -DTSD_TRACE variant build.

    tools/build_variant.sh trace "-DTSD_TRACE"
    TSDIFF_LIB=$PWD/tools/bin/lib_trace.so python tools/trace_combo.py [c2|c5small]

Prints, per role, the median / p90 duration of every phase (us, from s_memtime at 100 MHz or the shader clock --
the script calibrates against the launch's event time) and the per-CU occupancy picture."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tsdiff_amd import _lib, synth  # noqa: E402

_lib.LIB_PATH = os.environ.get("TSDIFF_LIB", os.path.join(ROOT, "tools", "bin", "lib_trace.so"))
from bench import make_models, to_dev  # noqa: E402
from tsdiff_amd.sampler import EnsembleSampler  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    fl_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 3   # filter layer of the traced launch (-1: node role only)
    h2 = (sys.argv[3] if len(sys.argv) > 3 else "h2") == "h2"  # split-f16 roles (default) or the fp32-MFMA ones
    dev = torch.device("cuda:0")
    lib = _lib.load()
    dbg = C.CDLL(_lib.LIB_PATH).tsd_debug_trace
    dbg.argtypes = [C.c_void_p]
    C.CDLL(_lib.LIB_PATH).tsd_debug_prec(1 if h2 else 0)
    cfg = synth.DEFAULT_MODEL_CONFIG
    H, L = 256, 7
    model = make_models(cfg, [0], dev)[0]
    if which == "c2":
        g = to_dev(synth.wb97xd3_like_batch(100, seed=1000), dev)
        g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
    elif which[0] == "g":   # gNNN: NNN graphs of the configs[1] distribution (g800: the tile counts of an 8-checkpoint ensemble)
        g = to_dev(synth.wb97xd3_like_batch(int(which[1:]), seed=1000), dev)
        g["pos"] = torch.randn(g["pos"].shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)) * 1.5
    else:
        g = to_dev(synth.dense_stress_batch(64, n=64, seed=1000), dev)
    s = EnsembleSampler([model])
    with torch.no_grad():
        s(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
    db = s._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
    N, PU = db.N, db.P // 2
    ea = torch.randn(max(PU, 1), H, device=dev)
    if h2:  # (the split-f16 filter role takes plane rows)
        ea16 = torch.empty_like(ea)
        _lib.check(lib.tsd_attr_planes(H, ea.shape[0], _lib.ptr(ea), _lib.ptr(ea16), None, _lib.stream_ptr()))
        ea = ea16
    wf = torch.randn(2, max(PU, 1), H, device=dev)
    xa, xb = torch.randn(N, H, device=dev), torch.empty(N, H, device=dev)
    hbuf = torch.randn(N, H, device=dev)
    node_tiles = (N + 15) // 16
    ftiles = (PU + 31) // 32
    grid = node_tiles + (ftiles if fl_arg >= 0 else 0)
    trace = torch.zeros(grid * 32, dtype=torch.int64, device=dev)

    def blk(layer, fl):
        _lib.check(lib.tsd_interaction_block(
            C.byref(db.cfg), _lib.ptr(db.weights16[0] if h2 else db.weights[0]), layer, N, db.enc.struct(), _lib.ptr(wf[0]), _lib.ptr(xa),
            _lib.ptr(hbuf), _lib.ptr(xb), fl, PU, db.enc_u.struct(), _lib.ptr(ea), _lib.ptr(wf[1]), _lib.stream_ptr()))
    for _ in range(5):
        blk(2, fl_arg)
    torch.cuda.synchronize()
    assert dbg(C.c_void_p(trace.data_ptr())) == 0
    blk(2, fl_arg)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    blk(2, fl_arg)
    ev1.record()
    torch.cuda.synchronize()
    dbg(C.c_void_p(0))
    # the same launch 50 times back to back (no tracing): average time per launch against the traced span of one
    for _ in range(3):
        blk(2, fl_arg)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(50):
        blk(2, fl_arg)
    ev1.record()
    torch.cuda.synchronize()
    print(f"50 untraced launches back to back: {ev0.elapsed_time(ev1) * 1e3 / 50:.2f} us per launch")
    t = trace.cpu().numpy().reshape(grid, 32).astype(np.int64)
    role = (t[:, 31] >> 40) & 255
    hw = t[:, 31] & 0xFFFFFFFF
    xcc = (t[:, 31] >> 32) & 15
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 12) & 1) << 7) | (xcc << 8)  # cu_id, se_id, sh_id, xcc
    # s_memtime counters are per XCD (unsynchronised bases): align every XCD at its first workgroup's start
    for x in range(16):
        m = (xcc == x) & (role > 0)
        if m.any():
            t[m, :31] -= t[m, 0].min()
    t0 = 0
    ev_us = 0.0
    span = max(t[role == r][:, 6 if r == 2 else 7].max() for r in (1, 2) if (role == r).any())
    tick_us, tn = 1.0 / 2400.0, "assumed 2.4 GHz shader clock"
    print(f"longest traced XCD span {span} ticks = {span * tick_us:.1f} us ({tn})")
    names = {1: ["start", "aggregated", "gemm lin2", "epi+bar", "gemm lin", "epi+bar", "gemm lin1", "stored"],
             2: ["start", "A tile in LDS", "gemm nn0", "ssp+bar", "gemm nn2", "epi+bar", "stored"]}
    for r, rn in ((1, "node role"), (2, "filter role")):
        m = role == r
        if not m.any():
            continue
        tt = (t[m][:, : len(names[r])] - t0) * tick_us
        print(f"{rn}: {m.sum()} workgroups; start median {np.median(tt[:, 0]):.1f} us (max {tt[:, 0].max():.1f}), "
              f"end median {np.median(tt[:, -1]):.1f} (max {tt[:, -1].max():.1f})")
        d = np.diff(tt, axis=1)
        for k in range(d.shape[1]):
            print(f"    {names[r][k + 1]:16s} median {np.median(d[:, k]):6.2f}  p90 {np.percentile(d[:, k], 90):6.2f}  max {d[:, k].max():6.2f} us")
    m = role == 2
    if m.any():
        w1 = (t[m][:, 8:16] - t[m][:, 1:2]) * tick_us   # every wave's GEMM-1 end relative to the post-load barrier
        w2 = (t[m][:, 16:24] - t[m][:, 3:4]) * tick_us
        print(f"filter role, per-wave GEMM ends after the barrier: nn0 first {np.median(w1.min(1)):.2f} last {np.median(w1.max(1)):.2f} us;"
              f" nn2 first {np.median(w2.min(1)):.2f} last {np.median(w2.max(1)):.2f} us (medians over workgroups)")
    m = role == 1
    if m.any():
        wa = (t[m][:, 16:24] - t[m][:, 0:1]) * tick_us
        print(f"node role, per-wave aggregation end after start: first {np.median(wa.min(1)):.2f} last {np.median(wa.max(1)):.2f} us")
    # launch-relative picture: the clocks are not synchronised across the chip, so every CU is aligned at the start of
    # its first workgroup (all CUs receive one within the first microsecond of a launch)
    live = role > 0
    rel_s, rel_e = np.zeros(grid), np.zeros(grid)
    for c in set(cu[live].tolist()):
        m = live & (cu == c)
        base = t[m, 0].min()
        rel_s[m] = (t[m, 0] - base) * tick_us
        rel_e[m] = (np.where(role[m] == 2, t[m, 6], t[m, 7]) - base) * tick_us
    for r, rn in ((1, "node"), (2, "filter")):
        m = role == r
        if m.any():
            print(f"launch-relative ({rn}): starts median {np.median(rel_s[m]):.1f} p95 {np.percentile(rel_s[m], 95):.1f} us; "
                  f"ends median {np.median(rel_e[m]):.1f}  p95 {np.percentile(rel_e[m], 95):.1f}  max {rel_e[m].max():.1f} us")
    # per-CU picture
    cus = {}
    for i in range(grid):
        if role[i]:
            cus.setdefault(int(cu[i]), []).append(int(role[i]))
    kinds = {}
    for c, v in cus.items():
        k = "".join(sorted("NF"[x - 1] for x in v))
        kinds[k] = kinds.get(k, 0) + 1
    print(f"{len(cus)} CUs used; workgroups per CU by kind: {dict(sorted(kinds.items()))}")
    for k in sorted(kinds):
        ends = []
        for c, v in cus.items():
            if "".join(sorted("NF"[x - 1] for x in v)) == k:
                idx = [i for i in range(grid) if role[i] and int(cu[i]) == c]
                ends.append(max((t[i, 6 if role[i] == 2 else 7] - t0) * tick_us for i in idx))
        print(f"    CUs with {k}: last workgroup ends at median {np.median(ends):.1f} us, max {max(ends):.1f}")


if __name__ == "__main__":
    main()
