#!/usr/bin/env python3
"""A/B of the fused training step's arithmetic (OPTIONS.train_gemm): gradients of ONE step in "h2" against "f32" from the
same parameters, batch, time steps and noise (per tensor: max|d| / max|ref|), then the step time of both.

usage: python tools/train_ab.py [--graphs 200] [--steps 30]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=200)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    import bench
    from tsdiff_amd import synth
    from tsdiff_amd.options import OPTIONS
    dev = torch.device("cuda:0")
    model = bench.make_models(synth.DEFAULT_MODEL_CONFIG, range(1), dev)[0]
    g = bench.to_dev(synth.wb97xd3_like_batch(a.graphs, seed=2000), dev)
    g["pos"] = (g["pos"] * 1.5).contiguous()
    G = a.graphs
    gen = torch.Generator(device="cpu").manual_seed(5)
    ts = torch.randint(0, 5000, (G,), generator=gen).to(dev)
    noise = torch.randn(g["pos"].shape, generator=gen).to(dev)
    model.train()
    out = {}
    for mode in ("f32", "h2"):
        OPTIONS.train_gemm = mode
        model._train_f32 = False
        model.zero_grad(set_to_none=True)
        loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                              g["batch"], g["num_nodes_per_graph"], G, _time_step=ts, _pos_noise=noise)
        loss.mean().backward()
        torch.cuda.synchronize()
        out[mode] = (loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        print(mode, "loss mean", float(loss.mean()), "fell back to f32" if getattr(model, "_train_f32", False) else "")
    lf, gf = out["f32"]
    lh, gh = out["h2"]
    print("loss: max|d| / max|ref| = %.3e" % float((lh - lf).abs().max() / lf.abs().max()))
    worst = []
    for k in gf:
        s = float(gf[k].abs().max())
        e = float((gh[k] - gf[k]).abs().max())
        worst.append((e / max(s, 1e-30), k, s))
    worst.sort(reverse=True)
    for r, k, s in worst[:12]:
        print("  %-52s rel %.3e  (scale %.3e)" % (k, r, s))
    if a.no_time:
        return
    for mode in ("f32", "h2", "f32", "h2"):
        OPTIONS.train_gemm = mode
        model._train_f32 = False
        dt, last, N, _ = bench.run_train(model, G, a.steps, 5, False, dev, 0, None)
        print("%s: %.3f ms/step (loss %.4g, %d atoms)" % (mode, dt / a.steps * 1e3, last, N))


if __name__ == "__main__":
    main()
