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
    monkeypatch.setattr(engine, "FUSED_STEP_TAIL", bool(on))


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
