#!/usr/bin/env python3
"""One data-parallel training step with a VISIBLE stand-in for the gradient all-reduce, for a kernel timeline on a one-GPU
lease (a one-rank RCCL all-reduce is a no-op: it launches nothing a kernel trace could show):

    rocprofv3 --kernel-trace -d /tmp/p -o dp -- python3 tools/dp_overlap_timeline.py
    python tools/step_timeline.py /tmp/p/.../dp_results.db pair_count > profiles/r04_train_timeline.md

`reduce_fn` = an in-place  x *= 1  over the tensor it is handed (an elementwise kernel of the all-reduce's byte count on
the stream the collective would run on).  tsdiff_amd.distributed.dp_backward hands it the interaction blocks' gradients
(9.2 of the 11.1 MB) on a side stream as soon as the library's event says they are final, then head and tail of the flat
vector after the backward pass."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_models, to_dev  # noqa: E402
from tsdiff_amd import optim, synth  # noqa: E402
from tsdiff_amd.distributed import dp_backward  # noqa: E402

dev = torch.device("cuda:0")
cfg = synth.DEFAULT_MODEL_CONFIG
model = make_models(cfg, [0], dev)[0]
model.train()
from tsdiff_amd.utils import AttrDict  # noqa: E402
opt = optim.get_optimizer(AttrDict({"type": "adam", "lr": 5e-4, "weight_decay": 0.0, "beta1": 0.95, "beta2": 0.999}), model)
batches = [to_dev(synth.wb97xd3_like_batch(200, seed=50 + k), dev) for k in range(3)]
calls = []


def stand_in(t):
    calls.append(t.numel())
    t.mul_(1.0)


for step in range(6):
    g = batches[step % 3]
    model._batches.clear()
    opt.zero_grad()
    loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
                          g["num_nodes_per_graph"], 200)
    dp_backward(model, loss, reduce_fn=stand_in, overlap=True)  # (the three-range form is opt-in since round 6)
    optim.clip_grad_norm_(model.parameters(), 3000.0)
    opt.step()
torch.cuda.synchronize()
print("reduce calls of the last step (floats):", calls[-4:], "->", model._last_reduce)
