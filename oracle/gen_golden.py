"""TEST INFRASTRUCTURE ONLY -- golden-vector generator (build container only).

Runs the UNCHANGED reference sources from /root/reference (imported through
oracle/ref_shims.py) on torch-CPU and writes small input/output fixtures to
tests/golden/*.npz.  No reference source text is written anywhere: the fixtures
hold numeric inputs and the reference's numeric outputs only.

    python oracle/gen_golden.py            # regenerates every fixture

Weights: the trained checkpoints are LFS blobs missing from the reference tree
(.MISSING_LARGE_BLOBS:3-12), so the reference model is loaded with the
closed-form synthetic state dict of tsdiff_amd/synth.py (`load_state_dict`),
which tests regenerate from (config, seed) instead of shipping an 11 MB blob.

Real graph: birkholz_benchmark/rxn_0/samples_all.pkl (13-atom reaction; the
only featurised reaction in the tree) is read with a stub unpickler and its
tensors are stored verbatim as fixture `rxn0_graph.npz`.
"""
import contextlib
import io
import json
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_shims  # noqa: E402
from tsdiff_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_rxn0():
    class Stub:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, s):
            if isinstance(s, dict):
                self.__dict__.update(s)

    class U(pickle.Unpickler):
        def find_class(self, mod, name):
            if mod.split(".")[0] in ("torch_geometric", "rdkit", "networkx"):
                return type(name, (Stub,), {})
            return super().find_class(mod, name)

    with open(os.path.join(ref_shims.REFERENCE_ROOT, "birkholz_benchmark/rxn_0/samples_all.pkl"), "rb") as f:
        data = U(f).load()
    d0 = data[0].__dict__
    g = {
        "atom_type": d0["atom_type"].numpy(),
        "r_feat": d0["r_feat"].numpy(),
        "p_feat": d0["p_feat"].numpy(),
        "bond_index": d0["edge_index"].numpy(),
        "bond_type": d0["edge_type"].numpy(),
        "pos_gen": np.stack([data[k].__dict__["pos_gen"].numpy() for k in range(8)]),
    }
    return g


def build_reference_model(R, cfg_dict, seed):
    cfg = R.EasyDict(cfg_dict)
    torch.manual_seed(1234)
    model = R.get_model(cfg)
    sd = synth.synth_state_dict(cfg_dict, seed)
    full = model.state_dict()
    for k, v in sd.items():
        assert tuple(full[k].shape) == tuple(v.shape), (k, full[k].shape, v.shape)
        full[k] = torch.from_numpy(v)
    # aliases model.* / model_embedding.* share storage with the named modules
    model.load_state_dict({k: full[k] for k in full if not k.startswith("model")}, strict=False)
    for k, v in sd.items():
        assert torch.equal(model.state_dict()[k], torch.from_numpy(v)), k
    model.eval()
    return model, cfg


def tt(batch):
    return {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in batch.items()}


def run_forward_with_trace(R, model, b):
    """forward + intermediates captured with forward hooks (no reference code touched)."""
    trace = {}
    enc = model.encoder
    hooks = []
    hooks.append(enc.interactions[0].conv.register_forward_hook(
        lambda m, i, o: trace.__setitem__("conv0_out", o.detach().clone())))
    for l, blk in enumerate(enc.interactions):
        hooks.append(blk.register_forward_hook(
            lambda m, i, o, l=l: trace.__setitem__(f"block{l}_out", o.detach().clone())))
    hooks.append(enc.register_forward_hook(lambda m, i, o: trace.__setitem__("h_final", o.detach().clone())))
    calls = []
    orig = model._extend_condensed_graph_edge

    def spy(*a, **k):
        out = orig(*a, **k)
        calls.append(out)
        return out

    model._extend_condensed_graph_edge = spy
    emb_calls = []
    orig_e = model._condensed_edge_embedding

    def spy_e(*a, **k):
        out = orig_e(*a, **k)
        emb_calls.append(out.detach().clone())
        return out

    model._condensed_edge_embedding = spy_e
    with torch.no_grad():
        G = int(b["num_graphs"])
        t = torch.zeros(G, dtype=torch.long)
        edge_inv, edge_index, edge_length = model.forward(
            b["atom_type"], b["r_feat"], b["p_feat"], b["pos"], b["bond_index"], b["bond_type"],
            b["batch"], t, return_edges=True)
        node_eq = R.geometry.eq_transform(edge_inv, b["pos"], edge_index, edge_length)
    for h in hooks:
        h.remove()
    del model._extend_condensed_graph_edge
    del model._condensed_edge_embedding
    res = {
        "edge_inv": edge_inv, "edge_index": edge_index, "edge_length": edge_length, "node_eq": node_eq,
        "enc_edge_index": calls[0][0], "enc_type_r": calls[0][2], "enc_type_p": calls[0][3],
        "out_type_r": calls[-1][2], "out_type_p": calls[-1][3],
        "enc_edge_attr": emb_calls[0], "out_edge_attr": emb_calls[-1],
        "h_final": trace["h_final"], "conv0_out": trace["conv0_out"],
    }
    return {k: v.numpy() for k, v in res.items()}


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]


def save(name, meta, **arrays):
    if ONLY and name not in ONLY:
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def inputs_of(b):
    return {"in_" + k: np.asarray(v) for k, v in b.items() if k != "num_graphs"}


def main():
    R = ref_shims.import_reference()
    rxn0 = load_rxn0()
    save("rxn0_graph", {"source": "birkholz_benchmark/rxn_0/samples_all.pkl[0..7]"}, **rxn0)
    g0 = {k: rxn0[k] for k in ("atom_type", "r_feat", "p_feat", "bond_index", "bond_type")}

    full_cfg = synth.DEFAULT_MODEL_CONFIG
    small_cfg = synth.small_model_config(64, 2)
    model_full, _ = build_reference_model(R, full_cfg, seed=0)
    model_small, _ = build_reference_model(R, small_cfg, seed=1)

    # --- A: rxn0, B=1, full model, generated geometry ------------------------------------
    b = synth.replicate(g0, 1, [rxn0["pos_gen"][0]])
    out = run_forward_with_trace(R, model_full, tt(b))
    save("fwd_rxn0_b1_full", {"cfg": full_cfg, "seed": 0}, **inputs_of(b), **out)

    # --- B: rxn0, B=4, noisy geometries (E_enc != E_out, C=0 edges), full model -----------
    rng = np.random.default_rng(7)
    poss = [rxn0["pos_gen"][1], (rng.standard_normal((13, 3)) * 12.17).astype(np.float32),
            (rng.standard_normal((13, 3)) * 4.4).astype(np.float32),
            (rng.standard_normal((13, 3)) * 7.36).astype(np.float32)]
    b = synth.replicate(g0, 4, poss)
    out = run_forward_with_trace(R, model_full, tt(b))
    out.pop("enc_edge_attr"); out.pop("out_edge_attr")  # keep the fixture small
    save("fwd_rxn0_b4_sigma_full", {"cfg": full_cfg, "seed": 0}, **inputs_of(b), **out)

    # --- C: synthetic wb97xd3-like graphs, small model, spread geometries -----------------
    b = synth.wb97xd3_like_batch(6, seed=3, n_lo=5, n_hi=23)
    b["pos"] = (b["pos"] * np.repeat(np.asarray([0.8, 2.0, 4.0, 6.0, 9.0, 12.0], dtype=np.float32),
                                     b["num_nodes_per_graph"])[:, None]).astype(np.float32)
    out = run_forward_with_trace(R, model_small, tt(b))
    save("fwd_synth_b6_small", {"cfg": small_cfg, "seed": 1}, **inputs_of(b), **out)

    # --- C2: same graphs, encoder.smooth_conv = True (cosine cutoff, schnet.py:92-96) ------------------------
    import copy
    smooth_cfg = copy.deepcopy(small_cfg)
    smooth_cfg["encoder"]["smooth_conv"] = True
    if not ONLY or "fwd_synth_b6_small_smooth" in ONLY:
        model_smooth, _ = build_reference_model(R, smooth_cfg, seed=1)
        out = run_forward_with_trace(R, model_smooth, tt(b))
        save("fwd_synth_b6_small_smooth", {"cfg": smooth_cfg, "seed": 1}, **inputs_of(b),
             edge_inv=out["edge_inv"], edge_index=out["edge_index"], edge_length=out["edge_length"],
             h_final=out["h_final"], node_eq=out["node_eq"])

    # --- D: ensemble forward (M=2), small model ------------------------------------------
    model_small2, _ = build_reference_model(R, small_cfg, seed=2)
    ens = R.sampler.EnsembleSampler([model_small, model_small2])
    bt_ = tt(b)
    with torch.no_grad():
        e_inv, e_idx, e_len = ens(bt_["atom_type"], bt_["r_feat"], bt_["p_feat"], bt_["pos"],
                                  bt_["bond_index"], bt_["bond_type"], bt_["batch"],
                                  torch.zeros(6, dtype=torch.long))
    save("ens_synth_b6_small", {"cfg": small_cfg, "seeds": [1, 2]}, **inputs_of(b),
         edge_inv=e_inv.numpy(), edge_index=e_idx.numpy(), edge_length=e_len.numpy())

    # --- E: LD / DDPM trajectories with recorded noise (config C1) -------------------------
    def run_sampler(models, b, n_steps, sampling_type, pos_init):
        bt = tt(b)
        noises = []
        orig = torch.randn_like

        def rec(x, *a, **k):
            n = orig(x, *a, **k)
            noises.append(n.clone())
            return n

        ens = R.sampler.EnsembleSampler(models)
        torch.manual_seed(2022)
        torch.randn_like = rec
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                pos, traj = ens.dynamic_sampling(
                    bt["atom_type"], bt["r_feat"], bt["p_feat"], torch.from_numpy(pos_init),
                    bt["bond_index"], bt["bond_type"], bt["batch"], int(b["num_graphs"]),
                    extend_order=True, n_steps=n_steps, step_lr=1e-7, clip=1000,
                    sampling_type=sampling_type)
        finally:
            torch.randn_like = orig
        return pos.numpy(), torch.stack(traj).numpy(), torch.stack(noises).numpy()

    b = synth.replicate(g0, 1, [np.zeros((13, 3), np.float32)])
    pos_init = np.random.default_rng(11).standard_normal((13, 3)).astype(np.float32)
    pos, traj, noises = run_sampler([model_full], b, 50, "ld", pos_init)
    save("ld_rxn0_b1_full_50", {"cfg": full_cfg, "seeds": [0], "n_steps": 50, "step_lr": 1e-7, "clip": 1000,
                                "sampling_type": "ld"},
         **inputs_of(b), pos_init=pos_init, noises=noises, pos_final=pos, traj=traj)

    b = synth.wb97xd3_like_batch(3, seed=5, n_lo=6, n_hi=12)
    pos_init = np.random.default_rng(12).standard_normal(b["pos"].shape).astype(np.float32)
    pos, traj, noises = run_sampler([model_small, model_small2], b, 20, "ld", pos_init)
    save("ld_synth_b3_small_ens2_20", {"cfg": small_cfg, "seeds": [1, 2], "n_steps": 20, "step_lr": 1e-7,
                                       "clip": 1000, "sampling_type": "ld"},
         **inputs_of(b), pos_init=pos_init, noises=noises, pos_final=pos, traj=traj)
    pos, traj, noises = run_sampler([model_small], b, 12, "ddpm", pos_init)
    save("ddpm_synth_b3_small_12", {"cfg": small_cfg, "seeds": [1], "n_steps": 12, "step_lr": 1e-7,
                                    "clip": 1000, "sampling_type": "ddpm"},
         **inputs_of(b), pos_init=pos_init, noises=noises, pos_final=pos, traj=traj)

    # --- E2: guess-TS modes of dynamic_sampling (sampler.py:149-177), small model -------------------
    def run_guess(models, b, pos_init, n_steps, **mode):
        bt = tt(b)
        noises, init = [], []
        o_like, o_randn = torch.randn_like, torch.randn

        def rec_like(x, *a, **k):
            n = o_like(x, *a, **k)
            noises.append(n.clone())
            return n

        def rec_randn(*a, **k):
            n = o_randn(*a, **k)
            init.append(n.clone())
            return n

        ens = R.sampler.EnsembleSampler(models)
        torch.manual_seed(77)
        torch.randn_like, torch.randn = rec_like, rec_randn
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                pos, traj = ens.dynamic_sampling(
                    bt["atom_type"], bt["r_feat"], bt["p_feat"], torch.from_numpy(pos_init), bt["bond_index"],
                    bt["bond_type"], bt["batch"], int(b["num_graphs"]), extend_order=True, n_steps=n_steps,
                    step_lr=1e-7, clip=1000, sampling_type="ld", **mode)
        finally:
            torch.randn_like, torch.randn = o_like, o_randn
        return pos.numpy(), torch.stack(traj).numpy(), torch.stack(noises).numpy(), (init[0].numpy() if init else None)

    b = synth.wb97xd3_like_batch(3, seed=5, n_lo=6, n_hi=12)
    guess = (b["pos"] * 1.2).astype(np.float32)
    pos, traj, noises, _ = run_guess([model_small], b, guess, 10, denoise_from_time_t=300)
    save("ld_guess_denoise_small", {"cfg": small_cfg, "seeds": [1], "n_steps": 10, "step_lr": 1e-7, "clip": 1000,
                                    "sampling_type": "ld", "denoise_from_time_t": 300},
         **inputs_of(b), pos_init=guess, noises=noises, pos_final=pos, traj=traj)
    pos, traj, noises, init = run_guess([model_small], b, guess, 10, denoise_from_time_t=300, noise_from_time_t=100)
    save("ld_guess_noise_denoise_small", {"cfg": small_cfg, "seeds": [1], "n_steps": 10, "step_lr": 1e-7,
                                          "clip": 1000, "sampling_type": "ld", "denoise_from_time_t": 300,
                                          "noise_from_time_t": 100},
         **inputs_of(b), pos_init=guess, noises=noises, init_noise=init, pos_final=pos, traj=traj)

    # --- E3: a checkpoint in the reference's on-disk format (train.py:221-231), small model -------------
    if not ONLY or "ckpt_small" in ONLY:
        full_cfg_tree = R.EasyDict({"model": small_cfg, "train": {"seed": 0, "batch_size": 200}})
        opt = torch.optim.Adam(model_small.parameters(), lr=5e-4, betas=(0.95, 0.999))
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.8, patience=10, min_lr=1.25e-4)
        os.makedirs(OUT, exist_ok=True)
        path = os.path.join(OUT, "ckpt_small.pt")
        torch.save({"config": full_cfg_tree, "model": model_small.state_dict(), "optimizer": opt.state_dict(),
                    "scheduler": sched.state_dict(), "iteration": 1000, "avg_val_loss": 1.25}, path)
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    # --- E4: legacy encoders called directly: GINEncoder, GaussianSmearingEdgeEncoder --
    if not ONLY or "legacy_gin_rbf" in ONLY:
        import importlib
        gin_mod = importlib.import_module("models.encoder.gin")
        edge_mod = importlib.import_module("models.encoder.edge")
        torch.manual_seed(5)
        Hh = 64
        enc = gin_mod.GINEncoder(hidden_dim=Hh, num_convs=3, activation="ReLU", embedding=True)
        b = synth.wb97xd3_like_batch(3, seed=11, n_lo=6, n_hi=12)
        ei = torch.from_numpy(b["bond_index"])
        zt = torch.from_numpy(b["atom_type"])
        ea = torch.randn(ei.shape[1], Hh)
        with torch.no_grad():
            out = enc(zt, ei, ea)
            conv0 = enc.convs[0](enc.node_emb(zt), ei, ea)
        # edge.py:28 references GaussianSmearing without importing it (NameError in the reference as shipped);
        # the class it means is models/encoder/schnet.py:14-23 -- bind that name, keep the reference's code
        edge_mod.GaussianSmearing = importlib.import_module("models.encoder.schnet").GaussianSmearing
        rbf = edge_mod.GaussianSmearingEdgeEncoder(num_gaussians=32, cutoff=10.0)
        dlen = (torch.rand(200, 1) * 22.0)
        typ = torch.randint(0, 26, (200,))
        with torch.no_grad():
            rout = rbf(dlen, typ)
        save("legacy_gin_rbf", {"hidden": Hh, "num_convs": 3},
             z=zt.numpy(), edge_index=ei.numpy(), edge_attr=ea.numpy(), gin_out=out.numpy(), conv0_out=conv0.numpy(),
             **{"gin." + k: v.numpy() for k, v in enc.state_dict().items()},
             rbf_d=dlen.numpy(), rbf_type=typ.numpy(), rbf_out=rout.numpy(),
             **{"rbf." + k: v.numpy() for k, v in rbf.state_dict().items()})


    # --- H: GeoDiff legacy dual-encoder network, the unchanged reference class -----------
    if not ONLY or any(n.startswith("dual_") for n in ONLY):
        import importlib
        dual_mod = importlib.import_module("models.epsnet.dualenc")

        def trainable(model):
            return {k: v for k, v in model.state_dict().items()
                    if not k.startswith("model_") and k not in ("betas", "alphas")}

        def dual_inputs(b, ts):
            bt = b["bond_type"] if ts else synth.single_bond_types(b["bond_type"])
            return {"atom_type": b["atom_type"], "bond_index": b["bond_index"], "bond_type": bt, "batch": b["batch"],
                    "num_nodes_per_graph": b["num_nodes_per_graph"], "pos": b["pos"]}

        def dual_fwd(model, x, pos, **kw):
            G = int(x["num_nodes_per_graph"].shape[0])
            with torch.no_grad():
                o = model(torch.from_numpy(x["atom_type"]), torch.from_numpy(pos), torch.from_numpy(x["bond_index"]),
                          torch.from_numpy(x["bond_type"]), torch.from_numpy(x["batch"]),
                          torch.zeros(G, dtype=torch.long), return_edges=True, **kw)
            names = ("edge_inv_global", "edge_inv_local", "edge_index", "edge_type", "edge_length", "local_edge_mask")
            return {n: v.numpy() for n, v in zip(names, o)}

        def dual_loss(model, x, G):
            cap = {}
            o_randint = torch.randint

            def rint(*a, **k):
                v = o_randint(*a, **k)
                cap.setdefault("half", v.clone())
                return v

            o_normal = torch.Tensor.normal_

            def nrm(self, *a, **k):
                o_normal(self, *a, **k)
                cap.setdefault("pos_noise", self.clone())
                return self

            # gin.py:139 adds the shortcut IN PLACE to the output of nn.ReLU; torch >= 1.9 differentiates ReLU
            # through its output and refuses (torch 1.8.1, which the reference pins, used the input).  Swap the
            # encoder's activation object for an input-differentiated ReLU -- same values, same gradient.
            class _ReLUFromInput(torch.nn.Module):
                def forward(self, t):
                    return t.clamp(min=0)

            saved_act = model.encoder_local.activation
            model.encoder_local.activation = _ReLUFromInput()
            torch.manual_seed(77)
            torch.randint, torch.Tensor.normal_ = rint, nrm
            try:
                model.zero_grad()
                loss, lg, ll = model.get_loss(
                    torch.from_numpy(x["atom_type"]), torch.from_numpy(x["pos"]), torch.from_numpy(x["bond_index"]),
                    torch.from_numpy(x["bond_type"]), torch.from_numpy(x["batch"]),
                    torch.from_numpy(x["num_nodes_per_graph"]), G, return_unreduced_loss=True)
            finally:
                torch.randint, torch.Tensor.normal_ = o_randint, o_normal
            loss.mean().backward()
            model.encoder_local.activation = saved_act
            T = model.num_timesteps
            ts_ = torch.cat([cap["half"], T - cap["half"] - 1])[:G]
            grads = {k: p.grad.numpy() for k, p in model.named_parameters()
                     if p.grad is not None and not k.startswith("model_")}
            return {"time_step": ts_.numpy(), "pos_noise": cap["pos_noise"].numpy(), "loss": loss.detach().numpy(),
                    "loss_global": lg.detach().numpy(), "loss_local": ll.detach().numpy()}, grads

        def dual_sample(model, x, G, n_steps, **kw):
            rec = []
            o_rl = torch.randn_like

            def rl(t, *a, **k):
                v = o_rl(t, *a, **k)
                rec.append(v.clone())
                return v

            torch.manual_seed(31)
            torch.randn_like = rl
            try:
                with contextlib.redirect_stderr(io.StringIO()):
                    pos, traj = model.langevin_dynamics_sample(
                        torch.from_numpy(x["atom_type"]), torch.from_numpy(x["pos"]),
                        torch.from_numpy(x["bond_index"]), torch.from_numpy(x["bond_type"]),
                        torch.from_numpy(x["batch"]), G, True, n_steps=n_steps, **kw)
            finally:
                torch.randn_like = o_rl
            return torch.stack(rec).numpy(), torch.stack(traj).numpy()

        for ts in (False, True):
            name = "dual_small_ts" if ts else "dual_small"
            if ONLY and name not in ONLY:
                continue
            cfg_d = synth.small_dual_config(ts=ts)
            torch.manual_seed(4321)
            model = dual_mod.DualEncoderEpsNetwork(R.EasyDict(cfg_d))
            with torch.no_grad():  # make nn.Embedding(max_norm=10) bind for half of the rows
                model.encoder_global.node_emb.weight[::2] *= 2.5
                for p in (model.encoder_local.node_emb.weight, model.edge_encoder_local.bond_emb.weight):
                    p *= 0.5
            sd0 = {k: v.clone().numpy() for k, v in trainable(model).items()}
            model.eval()
            b = synth.wb97xd3_like_batch(4, seed=21, n_lo=5, n_hi=12)
            x = dual_inputs(b, ts)
            G = 4
            arrays = {"sd." + k: v for k, v in sd0.items()}
            arrays.update({"in_" + k: v for k, v in x.items()})
            pos_far = (x["pos"] * 4.0).astype(np.float32)
            for tag, pos, kw in (("fwd", x["pos"], {}), ("far", pos_far, {}),
                                 ("noorder", pos_far, {"extend_order": False}),
                                 ("noradius", x["pos"], {"extend_radius": False})):
                arrays.update({f"{tag}.{k}": v for k, v in dual_fwd(model, x, pos, **kw).items()})
            lo, grads = dual_loss(model, x, G)
            arrays.update({"loss." + k: v for k, v in lo.items()})
            if not ts:  # every gradient in full; the TS variant keeps its size down with norms + three full ones
                arrays.update({"grad." + k: v for k, v in grads.items()})
            else:
                arrays.update({"grad." + k: grads[k] for k in ("edge_cat_global.0.weight", "edge_cat_local.2.weight",
                                                               "edge_encoder_local.bond_emb.weight")})
            grad_norms = {k: float(np.linalg.norm(v)) for k, v in grads.items()}
            if not ts:
                xs = dict(x)
                xs["pos"] = (x["pos"] * 0.02).astype(np.float32)  # pos_init; the sampler scales it by sigma_T
                for st, kw in (("ld", dict(sampling_type="ld", step_lr=1e-6)),
                               ("ddpm_noisy", dict(clip_local=3.0)),
                               ("ddpm_det", dict(sampling_type="ddpm_det", global_start_sigma=0.5, clip_pos=40.0)),
                               ("generalized", dict(sampling_type="generalized", eta=0.7, w_global=0.35))):
                    noise, traj = dual_sample(model, xs, G, 6, **kw)
                    arrays[f"samp.{st}.noise"], arrays[f"samp.{st}.traj"] = noise, traj
                arrays["samp.pos_init"] = xs["pos"]
            keys = sorted(k for k in model.state_dict().keys())
            save(name, {"cfg": cfg_d, "state_dict_keys": keys, "grad_norms": grad_norms}, **arrays)

        if not ONLY or "dual_qm9_fwd" in ONLY:  # the shipped legacy config, closed-form weights
            cfg_d = dict(synth.LEGACY_QM9_MODEL_CONFIG)
            torch.manual_seed(1)
            model = dual_mod.DualEncoderEpsNetwork(R.EasyDict(cfg_d))
            tr = trainable(model)
            sd = synth.hash_state_dict([(k, v.shape) for k, v in tr.items() if not k.endswith(".eps")], seed=3)
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
            model.eval()
            b = synth.wb97xd3_like_batch(3, seed=5, n_lo=8, n_hi=20)
            x = dual_inputs(b, False)
            o = dual_fwd(model, x, (x["pos"] * 2.0).astype(np.float32))
            save("dual_qm9_fwd", {"cfg": cfg_d, "seed": 3, "names": [k for k in sd]},
                 **{"in_" + k: v for k, v in x.items()}, pos=(x["pos"] * 2.0).astype(np.float32),
                 edge_inv_global=o["edge_inv_global"], edge_inv_local=o["edge_inv_local"], edge_type=o["edge_type"])

    # --- F: get_loss with captured random draws + gradient norms ---------------------------
    def run_loss(model, b, name, cfg, seed, all_grads=False):
        bt = tt(b)
        cap = {}
        o_randint, o_randn = torch.randint, torch.randn

        def rint(*a, **k):
            v = o_randint(*a, **k)
            cap.setdefault("half_1", v.clone())
            return v

        def rn(*a, **k):
            v = o_randn(*a, **k)
            cap.setdefault("pos_noise", v.clone())
            return v

        torch.manual_seed(99)
        torch.randint, torch.randn = rint, rn
        try:
            model.zero_grad()
            loss = model.get_loss(bt["atom_type"], bt["r_feat"], bt["p_feat"], bt["pos"], bt["bond_index"],
                                  bt["bond_type"], bt["batch"], bt["num_nodes_per_graph"],
                                  int(b["num_graphs"]))
        finally:
            torch.randint, torch.randn = o_randint, o_randn
        loss.mean().backward()
        G = int(b["num_graphs"])
        t0, t1 = 0, int(cfg["num_diffusion_timesteps"])
        half_1 = cap["half_1"]
        time_step = torch.cat([half_1, t0 + t1 - 1 - half_1])[:G]
        gn = {k: float(p.grad.norm()) for k, p in model.named_parameters()
              if p.grad is not None and not k.startswith("model")}
        if all_grads:  # every parameter's gradient, elementwise (small config: ~100 k floats)
            full = {"grad." + k: p.grad.numpy() for k, p in model.named_parameters()
                    if p.grad is not None and not k.startswith("model")}
            save(name, {"cfg": cfg, "seed": seed, "grad_norms": gn}, **inputs_of(b),
                 time_step=time_step.numpy(), pos_noise=cap["pos_noise"].numpy(), loss=loss.detach().numpy(), **full)
            return
        save(name, {"cfg": cfg, "seed": seed, "grad_norms": gn}, **inputs_of(b),
             time_step=time_step.numpy(), pos_noise=cap["pos_noise"].numpy(), loss=loss.detach().numpy(),
             grad_lin1_0=model.encoder.interactions[0].conv.lin1.weight.grad.numpy(),
             grad_out_w2=model.grad_dist_mlp.layers[2].weight.grad.numpy())

    b = synth.wb97xd3_like_batch(4, seed=8, n_lo=6, n_hi=14)
    b["pos"] = (b["pos"] * 1.5).astype(np.float32)
    run_loss(model_small, b, "loss_synth_b4_small", small_cfg, 1)
    run_loss(model_small, b, "grads_synth_b4_small", small_cfg, 1, all_grads=True)  # same draws (seed 99), all gradients
    b = synth.replicate(g0, 2, [rxn0["pos_gen"][2], rxn0["pos_gen"][3]])
    run_loss(model_full, b, "loss_rxn0_b2_full", full_cfg, 0)

    # --- F2: result-pickle fixture: the first two records of the reference's own samples_all.pkl (a data file the
    # reference tree holds), re-written record for record by tsdiff_amd.io (rdkit Mol blobs pass through as bytes)
    if not ONLY or "samples_rxn0_2" in ONLY:
        from tsdiff_amd import io as tio
        recs = tio.load_samples(os.path.join(ref_shims.REFERENCE_ROOT, "birkholz_benchmark/rxn_0/samples_all.pkl"))
        path = os.path.join(OUT, "samples_rxn0_2.pkl")
        tio.save_samples(path, recs[:2])
        print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")

    # --- G: schedule constants -----------------------------------------------------------
    save("schedule_full", {"cfg": full_cfg}, betas=model_full.betas.detach().numpy(),
         alphas=model_full.alphas.detach().numpy())


if __name__ == "__main__":
    main()
