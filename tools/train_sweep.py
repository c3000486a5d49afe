"""Split-f16 training step against the fp32-MFMA step over checkpoints, batch sizes and geometries: per case the worst\nparameter-gradient deviation (max|d| / max|ref| per tensor).  python tools/train_sweep.py"""
import sys, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tsdiff_amd import synth
from tsdiff_amd.options import OPTIONS
dev = torch.device('cuda:0')
worst_all = 0
for seed in (0, 1, 2):
    model = bench.make_models(synth.DEFAULT_MODEL_CONFIG, [seed], dev)[0]
    model.train()
    for kind, G in (("wb", 1), ("wb", 3), ("wb", 50), ("wb", 400), ("dense", 6)):
        b = synth.wb97xd3_like_batch(G, seed=10 * seed + G) if kind == "wb" else synth.dense_stress_batch(G, n=64, seed=seed)
        g = bench.to_dev(b, dev)
        if kind == "wb": g["pos"] = (g["pos"] * (0.7 + 0.6 * seed)).contiguous()
        gen = torch.Generator().manual_seed(seed)
        ts = torch.randint(0, 5000, (G,), generator=gen).to(dev)
        pn = torch.randn(g["pos"].shape, generator=gen).to(dev)
        out = {}
        for mode in ("f32", "h2"):
            OPTIONS.train_gemm = mode
            model._train_f32 = False
            model.zero_grad(set_to_none=True)
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], g["num_nodes_per_graph"], G, _time_step=ts, _pos_noise=pn)
            loss.mean().backward()
            torch.cuda.synchronize()
            out[mode] = (loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}, getattr(model, "_train_f32", False))
        lf, gf, _ = out["f32"]; lh, gh, fb = out["h2"]
        worst = max(float((gh[k] - gf[k]).abs().max()) / max(float(gf[k].abs().max()), 1e-30) for k in gf)
        fin = all(bool(torch.isfinite(v).all()) for v in gh.values())
        worst_all = max(worst_all, worst)
        print(f"seed {seed} {kind:5s} G={G:4d} N={g['pos'].shape[0]:6d}  loss rel {float((lh-lf).abs().max()/lf.abs().max()):.2e}  worst grad rel {worst:.2e}  finite {fin}  fell back {fb}")
print("worst overall", worst_all)
