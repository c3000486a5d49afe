"""Round-3 GPU tests: the fused step tail (one launch instead of step_post + scan + pair_fill), deeper
parity cases of VERDICT r02 (50 LD steps at batch 100, batch-200 training step, capacity guard) and the
single-GPU proofs of the multi-process entry points.  Everything goes through the C ABI (libtsdiff_hip.so)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tsdiff_amd.options import OPTIONS

from tests.test_gpu_parity import _dense_batch, _sample, _sampling_setup, make_model, to_dev

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _lists_snapshot(db):
    """every array the geometry build produces, cut to its valid length (host copies)"""
    snap = {}
    for name in ("enc", "out", "enc_u", "out_u", "diff_u"):
        el = getattr(db, name)
        E = el.num_edges()
        snap[name + ".count"] = E
        snap[name + ".row_ptr"] = el.row_ptr.cpu().numpy().copy()
        fields = ("dist", "type_r", "type_p") if name == "diff_u" else ("src", "dst", "dist", "type_r", "type_p", "pair_id")
        for f in fields:
            snap[f"{name}.{f}"] = getattr(el, f)[:E].cpu().numpy().copy()
    Eo = snap["out_u.count"]
    snap["attr_row"] = db.attr_row[:Eo].cpu().numpy().copy()
    snap["pair2out"] = db.pair2out[: db.P].cpu().numpy().copy()
    # pair2u is defined for src < dst pairs only (the other entries are never written)
    pid_u = np.concatenate([snap["enc_u.pair_id"], snap["out_u.pair_id"] + db.P]) if db.P else np.zeros(0, np.int64)
    snap["pair2u@members"] = db.pair2u.cpu().numpy()[pid_u].copy()
    return snap


def _set_tail(monkeypatch, on):
    from tsdiff_amd import engine
    monkeypatch.setattr(engine.OPTIONS, "fused_step_tail", bool(on))


@pytest.mark.parametrize("case", ["small", "ensemble2_ddpm", "sigma_far", "dense64"])
def test_fused_step_tail_equals_three_kernel_tail(case, dev, monkeypatch):
    """the one-launch step tail (update + next step's lists with the look-back scan) == step_post + scan + pair_fill,
    bit for bit: final positions, whole trajectory and every edge-list array after the last step; graph and eager"""
    from tsdiff_amd import synth
    from tsdiff_amd.sampler import EnsembleSampler
    kw = dict(sampling_type="ld")
    if case == "dense64":  # 64-atom graphs: the 256-thread instantiation, 63 chunks of pairs per graph
        cfg = synth.small_model_config(64, 2)
        ens = EnsembleSampler([make_model(cfg, 3, dev)])
        _, _, g = _dense_batch(5, 11, dev)
        G = 5
    elif case == "ensemble2_ddpm":
        ens, g, G = _sampling_setup(dev, graphs=9, seed=31, model_seeds=(4, 5))
        kw = dict(sampling_type="ddpm")
    else:
        ens, g, G = _sampling_setup(dev, graphs=70 if case == "small" else 7, seed=5)
    N = g["pos"].shape[0]
    n_steps = 6
    noises = torch.randn(n_steps, N, 3, device=dev)
    pos0 = g["pos"] * (30.0 if case == "sigma_far" else 1.0)  # far apart: radius membership differs per pair and step

    def run(on, use_graph):
        _set_tail(monkeypatch, on)
        pos, traj = ens.dynamic_sampling(g["atom_type"], g["r_feat"], g["p_feat"], pos0, g["bond_index"], g["bond_type"],
                                         g["batch"], G, True, n_steps=n_steps, step_lr=1e-7, clip=1000, noises=noises,
                                         use_graph=use_graph, **kw)
        db = ens._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
        return pos, torch.stack(traj), db

    ref_pos, ref_traj, _ = run(False, True)
    for use_graph in (True, False):
        pos, traj, db = run(True, use_graph)
        assert torch.equal(pos, ref_pos) and torch.equal(traj, ref_traj), (case, use_graph)
        # the lists the last tail launch left behind (those of the NEXT step: on the final positions) against the
        # stand-alone build (count + scan + fill) on the same positions
        lists = _lists_snapshot(db)
        db.geometry(pos)
        ref_lists = _lists_snapshot(db)
        assert lists.keys() == ref_lists.keys() and ref_lists["enc.count"] > 0
        for k in lists:
            assert np.array_equal(lists[k], ref_lists[k]), (case, use_graph, k)
    # the step counter of the state block ends on the last row of the step tables in both forms
    db = ens._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
    assert int(db.status[1]) == n_steps - 1 and int(db.status[0]) == 0


def test_fused_step_tail_edge_cases(dev, monkeypatch):
    """one-atom graphs (rows without pairs) at the start, middle and end of a batch, Philox draws, a second call on
    the same plan"""
    from tsdiff_amd import synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.small_model_config(64, 2)
    ens = EnsembleSampler([make_model(cfg, 8, dev)])
    def single_atom():
        z = np.zeros((1, 25), np.int64)
        z[0, [0, 2, 5, 9, 12, 16, 20, 23]] = 1
        return {"atom_type": np.asarray([6], np.int64), "r_feat": z, "p_feat": z.copy(),
                "pos": np.zeros((1, 3), np.float32), "bond_index": np.zeros((2, 0), np.int64),
                "bond_type": np.zeros(0, np.int64)}

    def split(b):  # a collated batch back into its graphs
        out, off = [], 0
        for n in b["num_nodes_per_graph"]:
            sel = (b["bond_index"][0] >= off) & (b["bond_index"][0] < off + n)
            out.append({"atom_type": b["atom_type"][off:off + n], "r_feat": b["r_feat"][off:off + n],
                        "p_feat": b["p_feat"][off:off + n], "pos": b["pos"][off:off + n],
                        "bond_index": b["bond_index"][:, sel] - off, "bond_type": b["bond_type"][sel]})
            off += n
        return out
    gs = split(synth.wb97xd3_like_batch(4, seed=3, n_lo=2, n_hi=9))
    # single-atom graphs (no bonds, no pairs: rows without pairs) first, in the middle and last
    b = synth.collate([single_atom(), gs[0], gs[1], single_atom(), gs[2], gs[3], single_atom()])
    g = to_dev({k: torch.from_numpy(np.asarray(v)) for k, v in b.items() if isinstance(v, np.ndarray)}, dev)
    G = int(len(b["num_nodes_per_graph"]))
    outs = []
    for on in (False, True):
        _set_tail(monkeypatch, on)
        a, ta = _sample(ens, g, G, 5, seed=77)
        c, tc = _sample(ens, g, G, 3, seed=78)  # the same plan again: ticket / records are re-armed per call
        outs.append((a, torch.stack(ta), c, torch.stack(tc)))
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x, y)


# ---------------------------------------------------------------------------------------------
# VERDICT r02 "test depth": configs[3] at its own size, a long trajectory at configs[1] size, the capacity guard
# ---------------------------------------------------------------------------------------------
def test_training_step_batch200_full_model(dev, monkeypatch):
    """BASELINE configs[3] at its own size (configs/train_config.yml: 200 graphs, H = 256, 7 blocks): the fused
    training step (size-switched launch shapes: batched wgrads, row-split counts) against the op-by-op autograd form
    on EVERY gradient, bit-reproducible on a rerun, and its loss against the pinned CPU oracle"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    from tests.test_gpu_parity import _grad_tol_check
    from tests.util import assert_close
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 200
    b = synth.wb97xd3_like_batch(G, seed=2000)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    t["pos"] = t["pos"] * 1.5
    g = to_dev(t, dev)
    gen = torch.Generator().manual_seed(11)
    ts = torch.randint(0, 5000, (G,), generator=gen)
    pn = torch.randn(t["pos"].shape, generator=gen)
    res = {}
    for mode in ("fused", "ops", "fused2"):
        monkeypatch.setattr(OPTIONS, "train", "ops" if mode == "ops" else "fused")
        model = make_model(cfg, 1, dev)
        model.train()
        loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                              g["batch"], g["num_nodes_per_graph"], G, _time_step=ts.to(dev), _pos_noise=pn.to(dev))
        loss.mean().backward()
        res[mode] = (loss.detach().cpu().numpy(),
                     {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None})
        del model
        torch.cuda.empty_cache()
    assert len(res["fused"][1]) == 7 + 9 * 7 + 6 + 4 and set(res["fused"][1]) == set(res["ops"][1])
    assert np.array_equal(res["fused"][0], res["fused2"][0])
    for k, v in res["fused"][1].items():
        assert np.array_equal(v, res["fused2"][1][k]), f"grad {k} not reproducible at batch 200"
        _grad_tol_check(v, res["ops"][1][k], k + " (fused vs op-by-op, batch 200)")
    assert_close(res["fused"][0], res["ops"][0], 2e-6, "loss fused vs op-by-op")
    o_loss = O.get_loss(O.to_torch_state(synth.synth_state_dict(cfg, 1)), cfg, t["atom_type"], t["r_feat"], t["p_feat"],
                        t["pos"], t["bond_index"], t["bond_type"], t["batch"], b["num_nodes_per_graph"], ts, pn)
    assert_close(res["fused"][0], o_loss.detach().numpy(), 5e-5, "loss vs oracle at batch 200")


def test_ld_50_steps_full_model_batch100_vs_oracle(dev):
    """BASELINE configs[1] over a LONG stretch: 50 LD steps of the 100-graph batch with the full model against the
    oracle's restatement of the reference loop (injected noise) -- the last 50 steps of the schedule, every step's
    positions.  (5e-5 of the coordinate scale: 50 steps of fp32 forwards at 1e-6 each, fed back through the update.)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    from tsdiff_amd.sampler import EnsembleSampler
    from tests.util import assert_close
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 0, dev)
    b = synth.wb97xd3_like_batch(100, seed=1000)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    g = to_dev(t, dev)
    N, K = t["pos"].shape[0], 50
    noises = torch.randn(K, N, 3, generator=torch.Generator().manual_seed(18))
    pos_init = torch.randn(N, 3, generator=torch.Generator().manual_seed(19)) * 1.5
    kw = dict(denoise_from_time_t=K)
    pos, traj = EnsembleSampler([model]).dynamic_sampling(
        g["atom_type"], g["r_feat"], g["p_feat"], pos_init.to(dev), g["bond_index"], g["bond_type"], g["batch"],
        100, True, n_steps=K, step_lr=1e-7, clip=1000, sampling_type="ld", noises=noises.to(dev), **kw)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    o_pos, o_traj = O.sample([O.to_torch_state(synth.synth_state_dict(cfg, 0))], cfg, t["atom_type"], t["r_feat"],
                             t["p_feat"], pos_init, t["bond_index"], t["bond_type"], t["batch"],
                             b["num_nodes_per_graph"], noises, K, **kw)
    assert_close(torch.stack(traj).numpy(), torch.stack(o_traj).numpy(), 5e-5, "50 LD steps at batch 100")
    assert_close(pos.cpu().numpy(), o_pos.numpy(), 5e-5, "final positions after 50 steps")


def _replicated_dense_batch(G, dev):
    """G 64-atom graphs replicated from 4 generated ones (only the sizes matter to the capacity tests; the first 8
    graphs are also evaluated on their own)"""
    from tsdiff_amd import synth
    b4 = synth.dense_stress_batch(4, n=64, seed=1)
    rep = G // 4
    rng = np.random.default_rng(3)
    pos = np.tile(b4["pos"], (rep, 1)) + rng.normal(0, 0.05, (G * 64, 3)).astype(np.float32)  # distinct geometries
    return {
        "atom_type": torch.from_numpy(np.tile(b4["atom_type"], rep)).to(dev),
        "r_feat": torch.from_numpy(np.tile(b4["r_feat"], (rep, 1))).to(dev),
        "p_feat": torch.from_numpy(np.tile(b4["p_feat"], (rep, 1))).to(dev),
        "pos": torch.from_numpy(pos).to(dev),
        "bond_index": torch.from_numpy(np.concatenate([b4["bond_index"] + 256 * r for r in range(rep)], axis=1)).to(dev),
        "bond_type": torch.from_numpy(np.tile(b4["bond_type"], rep)).to(dev),
        "batch": torch.arange(G, device=dev).repeat_interleave(64),
        "num_graphs": G,
    }


def test_capacity_guard_and_largest_accepted_batch(dev):
    """Capacity.  (1) a batch whose edge-attribute matrix would have >= 2^31 elements (2176 x 64-atom graphs) is
    refused up front with NotImplementedError / TSD_ERR_UNSUPPORTED -- by the Python host before any large
    allocation and by the C ABI's own guard.  (2) 2048 x 64-atom graphs (P H = 2.11e9, just below the limit and twice
    configs[4]) are accepted and RIGHT: the forward is deterministic, its first 8 graphs equal the 8-graph batch (whose
    shape is checked against the oracle in test_config_c5_shape_vs_oracle), edge_inv is bitwise symmetric."""
    import ctypes as C
    from tsdiff_amd import _lib, synth
    from tsdiff_amd.sampler import EnsembleSampler
    from tests.test_gpu_parity import run_forward
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 0, dev)
    lib = _lib.load()
    mc = model._cfg
    g = _replicated_dense_batch(2176, dev)
    with pytest.raises(NotImplementedError):
        EnsembleSampler([model]).dynamic_sampling(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"],
                                                  g["bond_type"], g["batch"], 2176, True, n_steps=1, sampling_type="ld")
    assert lib.tsd_forward_workspace_floats(C.byref(mc), 64 * 2176, 64 * 63 * 2176, 1) == 0  # the C ABI's own guard
    assert b"2^31" in lib.tsd_last_error()
    assert lib.tsd_forward_workspace_floats(C.byref(mc), 64 * 2048, 64 * 63 * 2048, 1) > 0
    del g
    G = 2048
    g = _replicated_dense_batch(G, dev)
    edge_inv, ei, el = run_forward(model, g, dev)
    E = G * 64 * 63
    assert ei.shape == (2, E) and bool(torch.isfinite(edge_inv).all())
    edge_inv2, _, _ = run_forward(model, g, dev)
    assert torch.equal(edge_inv, edge_inv2)
    del edge_inv2
    g8 = {k: (v[: 8 * 64] if k in ("atom_type", "r_feat", "p_feat", "pos", "batch") else v) for k, v in g.items()}
    nb8 = int((g["bond_index"][0] < 8 * 64).sum())
    g8["bond_index"], g8["bond_type"], g8["num_graphs"] = g["bond_index"][:, :nb8], g["bond_type"][:nb8], 8
    assert int(g8["bond_index"].max()) < 8 * 64
    inv8, ei8, _ = run_forward(model, g8, dev)
    E8 = 8 * 64 * 63
    assert torch.equal(ei8, ei[:, :E8])
    assert float((inv8 - edge_inv[:E8]).abs().max()) <= 2e-6 * float(inv8.abs().max())
    # the LAST graphs too (the highest row / tile indices of every kernel): equal to their own 8-graph batch
    lo = (G - 8) * 64
    gl = {k: (v[lo:] if k in ("atom_type", "r_feat", "p_feat", "pos") else v) for k, v in g.items()}
    sel = g["bond_index"][0] >= lo
    gl["bond_index"], gl["bond_type"] = g["bond_index"][:, sel] - lo, g["bond_type"][sel]
    gl["batch"], gl["num_graphs"] = g["batch"][lo:] - (G - 8), 8
    invl, eil, _ = run_forward(model, gl, dev)
    assert torch.equal(eil + lo, ei[:, E - E8:])
    assert float((invl - edge_inv[E - E8:]).abs().max()) <= 2e-6 * float(invl.abs().max())
    src, dst = ei[0], ei[1]
    jl, il = dst % 64, src % 64
    rev = dst * 63 + il - (il > jl).long()
    assert torch.equal(edge_inv.view(-1)[rev], edge_inv.view(-1))


def _param_grads(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad]).clone()


def test_prefetched_batch_gives_the_same_training_step(dev):
    """model.prefetch_batch builds the next batch's topology on a side stream during the current step (bench.py's
    training loop); get_loss on the prefetched batch == get_loss that builds it itself, bit for bit
    in the loss and in every parameter gradient, also when the GPU is busy on the main stream while it is built and
    when the cached batch is dropped right after the step (the allocator must not hand its memory out early)"""
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 0, dev)
    model.train()
    batches = [to_dev({k2: torch.from_numpy(v) for k2, v in synth.wb97xd3_like_batch(24, seed=900 + k).items()
                       if isinstance(v, np.ndarray)}, dev) for k in range(3)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    draws = []
    for g in batches:
        t = torch.randint(0, model.num_timesteps, (24,), device=dev, generator=gen)
        draws.append((t, torch.randn(g["pos"].shape, device=dev, generator=gen)))

    def topo(g):
        return (g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"], g["num_nodes_per_graph"])

    def run(prefetch):
        res = []
        model._batches.clear()
        busy = torch.randn(2048, 2048, device=dev)
        for k, g in enumerate(batches):
            model.zero_grad()
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                  g["batch"], g["num_nodes_per_graph"], 24, _time_step=draws[k][0], _pos_noise=draws[k][1])
            loss.mean().backward()
            model._batches.clear()
            if prefetch and k + 1 < len(batches):
                for _ in range(4):
                    busy = busy @ busy * 1e-3  # main-stream work in flight while the side stream builds
                db = model.prefetch_batch(*topo(batches[k + 1]))
                assert db.ready_event is not None and model._batches[0][2] is db
            res.append((loss.detach().clone(), _param_grads(model)))
        torch.cuda.synchronize()
        return res

    a, b = run(False), run(True)
    for (la, ga), (lb, gb) in zip(a, b):
        assert torch.equal(la, lb)
        assert torch.equal(ga, gb)
    # a second prefetch of the same tensors is a cache hit; a CPU model / tensor is a no-op
    model._batches.clear()
    d1 = model.prefetch_batch(*topo(batches[0]))
    assert model.prefetch_batch(*topo(batches[0])) is d1
    model._batches.clear()


def _random_reaction_batch(rng, G, n_lo, n_hi, scale):
    """G graphs with random bond graphs (0 bonds .. ring rich, R-only / P-only / both bonds of every order),
    1..n_hi atoms, positions at `scale` Angstrom: the topology zoo of test_random_topologies_edge_lists_bit_exact as
    full batches (types, features) the sampler can run on"""
    from tsdiff_amd import synth
    graphs = []
    for _ in range(G):
        n = int(rng.integers(n_lo, n_hi + 1))
        seen, bi, bt = set(), [], []
        for _ in range(int(rng.integers(0, 3 * n)) if n > 1 else 0):
            i, j = (int(v) for v in rng.integers(0, n, size=2))
            if i == j or (min(i, j), max(i, j)) in seen:
                continue
            seen.add((min(i, j), max(i, j)))
            r, p = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            if r == 0 and p == 0:
                r = 1
            bi += [(i, j), (j, i)]
            bt += [r * 22 + p] * 2
        ei = np.asarray(bi, dtype=np.int64).reshape(-1, 2)
        et = np.asarray(bt, dtype=np.int64)
        perm = np.lexsort((ei[:, 1], ei[:, 0])) if len(bi) else np.zeros(0, np.int64)
        graphs.append({"atom_type": rng.choice(np.asarray([1, 6, 7, 8, 9], dtype=np.int64), size=n),
                       "r_feat": synth._one_hot_feat(rng, n), "p_feat": synth._one_hot_feat(rng, n),
                       "pos": (rng.standard_normal((n, 3)) * scale).astype(np.float32),
                       "bond_index": ei[perm].T.copy() if len(bi) else np.zeros((2, 0), np.int64),
                       "bond_type": et[perm]})
    return synth.collate(graphs)


@pytest.mark.parametrize("hidden,convs,trials", [(64, 2, 24), (256, 3, 8)])
def test_sampling_paths_random_topologies_vs_oracle(hidden, convs, trials, dev, monkeypatch):
    """The sampling loop as shipped (one-launch step tail with the look-back scan, type-sorted embedding tiles, pair
    MLP inside the last block launch, folded edge_cat.2) on RANDOM topologies -- fragments without bonds, one-atom
    graphs, ring-rich graphs, every R/P bond-type combination, radius membership changing from step to step, edge
    orders 1..4, cutoffs from 'bonds only' to 'all pairs' -- against the pinned oracle's restatement of the
    reference loop (4 LD steps, injected noise, every step's positions), and against the same loop with the generic
    embedding kernel and the three-launch tail (same lists after the last step, positions within rounding)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import engine, synth
    from tsdiff_amd.sampler import EnsembleSampler
    from tests.util import assert_close
    rng = np.random.default_rng(4242 + hidden)
    n_typed = 0
    for trial in range(trials):
        G = int(rng.integers(1, 25))
        b = _random_reaction_batch(rng, G, 1, int(rng.choice([6, 20, 40])), float(rng.choice([0.7, 2.0, 5.0])))
        cfg = dict(synth.small_model_config(hidden, convs))
        cfg.update(edge_order=int(rng.integers(1, 5)), pred_edge_order=int(rng.integers(1, 5)),
                   edge_cutoff=float(rng.choice([0.0, 3.0, 10.0, 100.0])))
        t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        g = to_dev(t, dev)
        N = t["pos"].shape[0]
        n_steps = 4
        noises = torch.from_numpy(rng.standard_normal((n_steps, N, 3)).astype(np.float32))
        tag = f"trial {trial}: G={G} N={N} orders {cfg['edge_order']}/{cfg['pred_edge_order']} cutoff {cfg['edge_cutoff']}"

        def run(typed, tail):
            monkeypatch.setattr(engine.OPTIONS, "typed_tiles", typed)
            monkeypatch.setattr(engine.OPTIONS, "fused_step_tail", tail)
            model = make_model(cfg, 2, dev)
            ens = EnsembleSampler([model])
            pos, traj = ens.dynamic_sampling(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"],
                                             g["bond_type"], g["batch"], G, True, n_steps=n_steps, step_lr=1e-6, clip=1000,
                                             sampling_type="ld", noises=noises.to(dev))
            db = ens._bound_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"])
            return pos, torch.stack(traj), db

        pos, traj, db = run(True, True)
        n_typed += db.typed is not None
        lists = _lists_snapshot(db)
        o_pos, o_traj = O.sample([O.to_torch_state(synth.synth_state_dict(cfg, 2))], cfg, t["atom_type"], t["r_feat"],
                                 t["p_feat"], t["pos"], t["bond_index"], t["bond_type"], t["batch"],
                                 b["num_nodes_per_graph"], noises, n_steps, step_lr=1e-6)
        assert_close(traj.cpu().numpy(), torch.stack(o_traj).numpy(), 5e-5, tag + " vs oracle")
        # the lists left by the last tail launch == a stand-alone build on the final positions (bit exact)
        db.geometry(pos)
        ref_lists = _lists_snapshot(db)
        for k in lists:
            assert np.array_equal(lists[k], ref_lists[k]), (tag, k)
        pos_g, traj_g, _ = run(False, False)
        assert_close(traj.cpu().numpy(), traj_g.cpu().numpy(), 2e-5, tag + " typed tiles + fused tail vs generic kernels")
    assert n_typed >= trials // 2  # (batches without a single pair have no tiles: the generic kernel runs)


@pytest.mark.parametrize("hidden,convs,trials", [(64, 2, 6), (256, 2, 6)])
def test_training_step_random_topologies_fused_vs_op_by_op_and_oracle(hidden, convs, trials, dev, monkeypatch):
    """the fused training step on the random-topology zoo (batches without a separately embedded out edge, without
    any out edge, with one-atom graphs, with every R/P bond combination): loss and EVERY parameter gradient against
    the op-by-op autograd form of the primitive kernels, the loss against the pinned oracle (fp32 CPU)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    from tests.test_gpu_parity import _grad_tol_check
    from tests.util import assert_close
    rng = np.random.default_rng(777 + hidden)
    for trial in range(trials):
        G = int(rng.integers(1, 30))
        b = _random_reaction_batch(rng, G, 1 if trial % 2 else 2, int(rng.choice([5, 16, 40])), float(rng.choice([1.0, 2.5])))
        cfg = dict(synth.small_model_config(hidden, convs))
        cfg.update(edge_order=int(rng.integers(1, 5)), pred_edge_order=int(rng.integers(1, 5)),
                   edge_cutoff=float(rng.choice([0.0, 3.0, 10.0])))
        t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
        g = to_dev(t, dev)
        N = t["pos"].shape[0]
        ts = torch.from_numpy(rng.integers(0, 5000, size=G))
        pn = torch.from_numpy(rng.standard_normal((N, 3)).astype(np.float32))
        tag = f"trial {trial}: G={G} N={N} orders {cfg['edge_order']}/{cfg['pred_edge_order']} cutoff {cfg['edge_cutoff']}"
        res = {}
        for mode in ("fused", "ops"):
            monkeypatch.setattr(OPTIONS, "train", mode)
            model = make_model(cfg, 3, dev)
            model.train()
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                  g["batch"], g["num_nodes_per_graph"], G, _time_step=ts.to(dev), _pos_noise=pn.to(dev))
            loss.mean().backward()
            res[mode] = (loss.detach().cpu().numpy(),
                         {k: p.grad.cpu().numpy() for k, p in model.named_parameters() if p.grad is not None})
        assert set(res["fused"][1]) == set(res["ops"][1]), tag
        assert_close(res["fused"][0], res["ops"][0], 2e-6, tag + " loss fused vs op-by-op")
        for k, ref in res["ops"][1].items():
            _grad_tol_check(res["fused"][1][k], ref, f"{k} ({tag}, fused vs op-by-op)")
        sd = O.to_torch_state(synth.synth_state_dict(cfg, 3))
        o_loss = O.get_loss(sd, cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"], t["bond_type"],
                            t["batch"], b["num_nodes_per_graph"], ts, pn)
        assert_close(res["fused"][0], o_loss.detach().numpy(), 5e-5, tag + " loss vs oracle")


def test_diffuse_positions_equals_the_reference_expression(dev):
    """tsd_diffuse_positions (one launch) == condensenc.py:292-297 evaluated by torch op by op, bit for bit: the
    per-graph alphas and the perturbed positions, including graphs at both ends of the schedule"""
    from tsdiff_amd import _lib, synth
    from tsdiff_amd._lib import check, ptr, stream_ptr
    lib = _lib.load()
    model = make_model(synth.small_model_config(), 0, dev)
    alphas = model.alphas.detach()
    T = alphas.shape[0]
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    for G, n_max in ((1, 1), (7, 30), (200, 23)):
        nn = torch.randint(1, n_max + 1, (G,), generator=torch.Generator().manual_seed(G))
        batch = torch.repeat_interleave(torch.arange(G), nn).to(dev)
        N = int(nn.sum())
        ts = torch.randint(0, T, (G,), device=dev, generator=gen)
        ts[0], ts[-1] = 0, T - 1
        pos = torch.randn(N, 3, device=dev, generator=gen) * 3
        noise = torch.randn(N, 3, device=dev, generator=gen)
        out, a = torch.empty_like(pos), torch.empty(G, device=dev)
        check(lib.tsd_diffuse_positions(N, G, T, ptr(alphas), ptr(ts), ptr(batch), ptr(pos), ptr(noise), ptr(out), ptr(a),
                                        stream_ptr()))
        a_ref = alphas.index_select(0, ts)
        a_pos = a_ref.index_select(0, batch).unsqueeze(-1)
        ref = pos + noise * (1.0 - a_pos).sqrt() / a_pos.sqrt()
        assert torch.equal(a, a_ref)
        assert torch.equal(out, ref), float((out - ref).abs().max())
