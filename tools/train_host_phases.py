#!/usr/bin/env python3
"""Host time of the training loop's phases at batch 200 (bench.py's loop, no device sync inside): where the host waits.
    python tools/train_host_phases.py [graphs] [topology|pos|none]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tsdiff_amd import optim, synth
from tsdiff_amd.distributed import dp_backward
from types import SimpleNamespace
G = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "pos"
dev = torch.device("cuda:0")
model = bench.make_models(synth.DEFAULT_MODEL_CONFIG, [0], dev)[0]
model.train()
batches = []
for k in range(8):
    g = bench.to_dev(synth.wb97xd3_like_batch(G, seed=2000 + k), dev)
    g["pos"] = (g["pos"] * 1.5).contiguous()
    batches.append(g)
opt = optim.get_optimizer(SimpleNamespace(type="adam", lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
def topo(g):
    return (g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"], g["num_nodes_per_graph"])
import gc
acc = {}
def lap(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1
def step(i, rec):
    g = batches[i % 8]
    t = time.perf_counter()
    opt.zero_grad(); t = lap("zero_grad", t) if rec else t
    loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], g["num_nodes_per_graph"], G)
    t = lap("get_loss (forward)", t) if rec else t
    if mode == "pos":
        nxt = batches[(i + 1) % 8]
        model.prefetch_batch(*topo(nxt), pos=nxt["pos"], num_graphs=G)
        t = lap("prefetch", t) if rec else t
    dp_backward(model, loss); t = lap("dp_backward", t) if rec else t
    optim.clip_grad_norm_(model.parameters(), 3000.0); t = lap("clip", t) if rec else t
    opt.step(); t = lap("adam", t) if rec else t
    if mode == "pos":
        model._batches = model._batches[:1]
        return
    model._batches.clear()
    nxt = batches[(i + 1) % 8]
    if mode == "topology":
        model.prefetch_batch(*topo(nxt))
    t = lap("prefetch", t) if rec else t
gc.collect(); gc.freeze()
for i in range(30): step(i, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 300
for i in range(K): step(i, True)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"mode {mode}, {G} graphs: wall {wall / K * 1e3:.3f} ms/step; host time per phase (ms/step, waits included):")
for k, v in acc.items():
    print(f"   {k:22s} {v / K * 1e3:.3f}")
print(f"   {'sum':22s} {sum(acc.values()) / K * 1e3:.3f}")
