#!/usr/bin/env python3
"""A/B of the training loop's prefetch forms in ONE process (alternating blocks: the host's state is shared):
    python tools/ab_train_prefetch.py [graphs] [steps per block] [blocks]
modes: pos = the next batch prefetched right behind get_loss WITH its positions (prefetch_batch(pos=...): draws, diffusion and edge
lists of the next step built ahead, no host wait for the edge counts -- the default since round 6), pos-late = the same behind
opt.step(), early / late = the topology only (late = rounds 3-5), none = everything inside get_loss"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tsdiff_amd import synth
G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
model = bench.make_models(synth.DEFAULT_MODEL_CONFIG, [0], dev)[0]
res = {"pos": [], "pos-late": [], "early": [], "late": [], "none": []}
for b in range(B):
    for mode in ("pos", "pos-late", "early", "late", "none"):
        dt, last, N, _ = bench.run_train(model, G, K, 10, False, dev, 0, None, prefetch=False if mode == "none" else mode)
        res[mode].append(dt / K * 1e3)
for m, v in res.items():
    print(f"{m:9s} ms/step: " + " ".join(f"{x:.3f}" for x in v) + f"   median {np.median(v):.3f}  min {min(v):.3f}")
