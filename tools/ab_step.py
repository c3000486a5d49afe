#!/usr/bin/env python3
"""A/B timing of the whole sampling step (configs[1] by default) across library variants and host switches.

    python tools/ab_step.py [--workload c2|c5|ensN|gNNN|gNNNmM] [--steps 200] [--rounds 3] NAME=LIB[:tail0] ...

Every configuration runs in its own child process (one library per process), the configurations interleaved over
`rounds` rounds; prints ms/step min / median per configuration.  LIB = path of a libtsdiff_hip.so variant
(tools/build_variant.sh) or `default`; suffixes: `:tail0` runs the step tail as three launches, `:typed0` the generic embedding kernel, `:f32` the fp32-input MFMA forward, `:fused0` / `:fused1` without / always with the fused per-unit encoder."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(workload, steps, lib, flags):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from tsdiff_amd import _lib, engine, synth
    if lib != "default":
        _lib.LIB_PATH = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
    engine.OPTIONS.fused_step_tail = "tail0" not in flags
    engine.OPTIONS.typed_tiles = "typed0" not in flags
    if "perblock" in flags:
        engine.OPTIONS.one_launch = False  # the split-f16 forward as one launch per block
    if "f32" in flags:
        engine.OPTIONS.gemm = "f32"  # fp32-input MFMA instead of the split-f16 forward
    if "fused0" in flags:
        engine.OPTIONS.fused_encoder = False  # materialised filters (one launch per block) where the fused encoder applies
    if "fused1" in flags:
        engine.OPTIONS.fused_encoder = "force"  # the fused per-unit encoder also where the one-launch form applies
    from bench import SamplingRun, make_models, to_dev
    from tsdiff_amd.sampler import EnsembleSampler
    dev = torch.device("cuda:0")
    cfg = synth.DEFAULT_MODEL_CONFIG
    M = int(workload[3:]) if workload.startswith("ens") else 1  # ensN: N checkpoints on the configs[1] batch
    if workload[0] == "g" and "m" in workload:  # gNNNmM: NNN graphs, M checkpoints (g300m8: configs[2]'s per-GPU unit)
        workload, mm = workload.split("m")
        M = int(mm)
    models = make_models(cfg, range(M), dev)
    if workload == "c5":
        b = synth.dense_stress_batch(1024, n=64, seed=1000)
        G = 1024
    else:
        G = int(workload[1:]) if workload[0] == "g" else 100   # gNNN: NNN graphs of the configs[1] distribution
        b = synth.wb97xd3_like_batch(G, seed=1000)
    g = to_dev(b, dev)
    N = g["pos"].shape[0]
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    pos = g["pos"].clone() if workload == "c5" else torch.randn(N, 3, device=dev, generator=gen) * 1.5
    run = SamplingRun(EnsembleSampler(models), g, G, pos, True, 1234)
    run.run(5)
    t0 = time.perf_counter()
    while workload != "c5" and time.perf_counter() - t0 < 0.2:
        run.run(50)
    ts = []
    for _ in range(5):
        dt, p = run.timed(steps)
        ts.append(dt / steps * 1e3)
    assert torch.isfinite(p).all()
    print(f"RESULT {min(ts):.4f} {float(np.median(ts)):.4f} {float(p.double().abs().sum()):.6f}")


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], int(args[2]), args[3], args[4].split(","))
        return
    workload, steps, rounds, cfgs = "c2", 200, 3, []
    i = 0
    while i < len(args):
        if args[i] == "--workload":
            workload = args[i + 1]; i += 2
        elif args[i] == "--steps":
            steps = int(args[i + 1]); i += 2
        elif args[i] == "--rounds":
            rounds = int(args[i + 1]); i += 2
        else:
            name, spec = args[i].split("=", 1)
            parts = spec.split(":")
            cfgs.append((name, parts[0], ",".join(parts[1:]) or "-")); i += 1
    res = {c[0]: [] for c in cfgs}
    chk = {}
    for _ in range(rounds):
        for name, lib, flags in cfgs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", workload, str(steps), lib, flags],
                                 capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
            if not line:
                print(f"{name}: FAILED\n{out.stdout[-2000:]}\n{out.stderr[-3000:]}")
                continue
            mn, med, cs = line[0].split()[1:]
            res[name].append((float(mn), float(med)))
            chk[name] = cs
    print(f"workload {workload}, {steps} steps per timed call, 5 calls per process, {rounds} processes per configuration")
    for name, v in res.items():
        if v:
            print(f"  {name:24s} ms/step min {min(x[0] for x in v):.4f}  median-of-medians "
                  f"{sorted(x[1] for x in v)[len(v) // 2]:.4f}   checksum {chk[name]}")


if __name__ == "__main__":
    main()
