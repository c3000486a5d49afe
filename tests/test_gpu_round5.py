"""Round-5 GPU tests (through the C ABI, `-m gpu`): the transposed-accumulator tile GEMMs, the split planes' exactness, the
training-drift bound of the default arithmetic."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.test_gpu_parity import dev, make_model, run_forward, to_dev  # noqa: E402,F401
from tests.util import assert_close  # noqa: E402

RTOL = 1e-5  # north_star: eps within 1e-5 rel-fp32 of the reference

pytestmark = pytest.mark.gpu


def _stretched_batch(graphs, seed, dev):
    from tsdiff_amd import synth
    b = synth.wb97xd3_like_batch(graphs, seed=seed)
    b["pos"] = (b["pos"] * np.repeat(np.linspace(0.7, 9.0, graphs).astype(np.float32), b["num_nodes_per_graph"])[:, None])
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    return b, t, to_dev({**t, "num_graphs": graphs}, dev)


@pytest.mark.parametrize("graphs,seed", [(20, 5), (57, 11)])
def test_every_forward_form_gives_the_same_bits_on_stretched_batches(graphs, seed, dev, monkeypatch):
    """one-launch, launch-per-block and fused-encoder forwards of the split-f16 path on batches with > 10 A pairs, separately
    embedded out edges and partial tiles: bit-identical, and within 1e-5 of the oracle.  (Round 5 found the one-launch
    kernel's pair role 6e-5 off on 15 of 1791 pairs of the first batch: the compiler had fused ONE of the two uses of a
    float -> f16 conversion with the multiply before it -- split16.hpp split1 -- which no other test batch exposed.)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _stretched_batch(graphs, seed, dev)
    o_inv, o_ei, _ = O.forward(O.to_torch_state(synth.synth_state_dict(cfg, 3)), cfg, t["atom_type"], t["r_feat"], t["p_feat"],
                               t["pos"], t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    res = {}
    for form in ("one_launch", "per_block", "fused"):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", form == "one_launch")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", "force" if form == "fused" else False)
        model = make_model(cfg, 3, dev)
        inv, ei, _ = run_forward(model, g, dev)
        assert torch.equal(ei.cpu(), o_ei)
        assert_close(inv.cpu().numpy(), o_inv.numpy(), RTOL, f"edge_inv ({form})")
        res[form] = inv.clone()
    assert torch.equal(res["one_launch"], res["per_block"])
    assert torch.equal(res["fused"], res["per_block"])


def test_split_planes_reproduce_fp32_values_of_products(dev):
    """split16.hpp split1 on values that are PRODUCTS (swish outputs: x * sigmoid(x)) -- the case in which a fused
    multiply-convert would round the high plane differently from the value the low plane is computed against: the typed
    embedding's attribute rows feed the pair MLP through exactly such planes; the forward must stay within 2e-6 of an fp64
    evaluation of the oracle on a batch large enough to hit the rare double-rounding cases (1 in ~10^3 elements)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _stretched_batch(40, 21, dev)
    o64, _, _ = O.forward(O.to_torch_state(synth.synth_state_dict(cfg, 3), torch.float64), cfg, t["atom_type"], t["r_feat"],
                          t["p_feat"], t["pos"].double(), t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    model = make_model(cfg, 3, dev)
    inv, _, _ = run_forward(model, g, dev)
    err = float((inv.cpu().double() - o64).abs().max() / o64.abs().max())
    assert err <= 2e-6, f"split-f16 forward {err:.2e} from the fp64 evaluation"


def test_train_drift_h2_within_fp32_noise(dev):
    """300 optimizer steps at batch 200 from one initialisation over the same batches / time steps / noise in three
    arithmetics (tools/train_drift.py; table: profiles/r06_train_drift.md), plus two fp32 trainings whose initialisation is
    perturbed by ONE ULP.  After 50 steps the split-f16 training's parameter distance from the fp32 training is within 2 x
    the distance of a second fp32 training that differs by summation order only (op-by-op autograd form).  From step ~150
    on the training dynamics amplify ANY difference exponentially (x 2 per 50 steps; the one-ulp runs are 6e-6 .. 1.4e-5
    away after 300 steps with a x 2.3 spread between seeds), so late ratios of two such distances are samples of a broad
    distribution -- r05 measured h2 / ops = 2.74 at step 300, r06 1.05 with the same code path.  What is asserted at every
    snapshot instead: the split-f16 training is no further from the fp32 training than an fp32 training that started one
    ulp away, both five orders of magnitude below the distance the parameters travel; the loss curves agree to 1e-4"""
    from tools.train_drift import drift, perturbed_rows
    from tsdiff_amd import synth
    rows, trips, moved, res = drift(300, 200, 50, dev, synth.DEFAULT_MODEL_CONFIG, perturbed=2)
    pr = perturbed_rows(res)
    assert trips == 0
    assert moved > 1e-3, "the parameters did not move: the run proves nothing"
    for s, lf, lh, lo, dh, do in rows:
        print(f"step {s}: loss f32 {lf:.6g} h2 {lh:.6g} ops {lo:.6g}; h2 vs f32 {dh:.3e}, ops vs f32 {do:.3e}, one-ulp runs {pr[s]}")
        if s == 0:
            assert dh == 0.0 and do == 0.0
            continue
        assert np.isfinite(lh) and np.isfinite(lf)
        assert dh <= 1e-3 * moved and do <= 1e-3 * moved
        assert dh <= max(pr[s]) + 1e-7, f"step {s}: h2 {dh:.3e} beyond the one-ulp-perturbed fp32 trainings {pr[s]}"
        if s == 50:
            assert dh <= max(2.0 * do, max(pr[s])) + 1e-7, f"h2 {dh:.3e} vs fp32 re-association {do:.3e}"
    assert rows[-1][0] == 300
    lf, lh = np.array(res["f32"][0]), np.array(res["h2"][0])
    assert float(np.max(np.abs(lh - lf) / np.abs(lf))) < 1e-4


def test_fused_encoder_refuses_a_broken_unit_partition(dev, monkeypatch):
    """tsd_batch.unit_node is the caller's: a unit that cuts a graph (its local atom numbers would leave the unit's LDS rows)
    makes the workgroup report TSD_STATUS_INTERNAL instead of computing -- `force` raises, the default policy reruns the
    batch on the forms without units and returns the right answer"""
    from tests.test_gpu_round4 import _db
    from tsdiff_amd import _lib, engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 72  # (4320 atoms: past the one-launch form's size, so the default policy takes the fused encoder by itself)
    b = synth.dense_stress_batch(G, n=60, seed=3)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    g = to_dev({**t, "num_graphs": G}, dev)
    monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
    ref, ref_ei, _ = run_forward(make_model(cfg, 2, dev), g, dev)
    for mode in ("force", True):
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", mode)
        model = make_model(cfg, 2, dev)
        inv, ei, _ = run_forward(model, g, dev)
        assert torch.equal(inv, ref)
        db = _db(model)
        assert db.unit_node is not None and db.unit_node.numel() == G + 1
        db.unit_node[1] += 3  # units 0 and 1 now cut graph 1
        if mode == "force":
            with pytest.raises(_lib.TsdError):
                run_forward(model, g, dev)
        else:
            inv2, _, _ = run_forward(model, g, dev)
            assert torch.equal(inv2, ref) and db.per_block


@pytest.mark.parametrize("graphs,M", [(12, 4), (40, 2), (100, 2), (100, 4), (30, 3)])
def test_ensemble_in_the_one_launch_form_matches_the_block_launches(graphs, M, dev, monkeypatch):
    """An ensemble whose checkpoints make whole groups (as many per group as keep the node workgroups within half of the
    chip's slots) runs in the one-launch kernel, group after group in one grid (api.hip mega_shape / mega_group): one group
    (12 x 4, 40 x 2, 100 x 2), two groups of two (100 x 4); 30 x 3 fits one group as well.  Bit-identical to the
    launch-per-block forms, per checkpoint (edge_inv of every checkpoint), and to each checkpoint run alone."""
    from tests.test_gpu_round4 import _db
    from tsdiff_amd import engine, synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _stretched_batch(graphs, 7 + graphs + M, dev)
    args = (g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
    out = {}
    for form in (True, False):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", form)
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
        ens = EnsembleSampler([make_model(cfg, 10 + k, dev) for k in range(M)])
        with torch.no_grad():
            mean = ens(*args)[0].clone()
        db = ens._bound_batch(*args[:3], *args[4:7])
        assert not db.per_block
        out[form] = (mean, db.edge_inv_u.clone())
    assert torch.equal(out[True][1], out[False][1])   # every checkpoint's pair outputs
    assert torch.equal(out[True][0], out[False][0])
    # checkpoint k alone (the single-checkpoint one-launch form)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    E = out[True][0].shape[0]
    for k in (0, M - 1):
        one = EnsembleSampler([make_model(cfg, 10 + k, dev)])
        with torch.no_grad():
            one(*args)
        d1 = one._bound_batch(*args[:3], *args[4:7])
        assert torch.equal(d1.edge_inv_u[0, :E // 2], out[True][1][k, :E // 2]), k   # (undirected out edges)


def test_ensemble_one_launch_wait_that_gives_up_reruns_per_block(dev, monkeypatch):
    """the bounded waits of the one-launch kernel with an ENSEMBLE in it (two groups of two checkpoints at batch 100): fault
    injection (tsd_batch.reserved bit 3: the last filter tile of every checkpoint is never run) -> TSD_STATUS_INTERNAL, the
    forward is rerun on the block launches and returns their bits without raising"""
    from tests.test_gpu_round4 import _batch
    from tsdiff_amd import _lib, engine, synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(100, 1000, dev)
    args = (g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
    monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", False)
    models = [make_model(cfg, 20 + k, dev) for k in range(4)]
    with torch.no_grad():
        ref = EnsembleSampler(models)(*args)[0].clone()
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    ens = EnsembleSampler([make_model(cfg, 20 + k, dev) for k in range(4)])
    with torch.no_grad():
        ok = ens(*args)[0].clone()
    db = ens._bound_batch(*args[:3], *args[4:7])
    assert torch.equal(ok, ref) and not db.per_block
    db.test_flags = 8
    with torch.no_grad():
        out = ens(*args)[0].clone()
    assert db.per_block, "the fault was not noticed"
    assert int(db.status[0].item()) & (_lib.STATUS_INTERNAL | _lib.STATUS_RANGE) == 0
    assert torch.equal(out, ref)


@pytest.mark.parametrize("graphs,M", [(100, 1), (100, 2), (200, 1)])
def test_one_launch_kernel_with_32_and_64_row_tiles_gives_the_same_bits(graphs, M, dev, monkeypatch):
    """OPTIONS.wide_filter_tiles (tsd_batch.reserved bit 1) in the one-launch kernel: 64-row filter tiles from 384 32-row tiles
    per block (x checkpoints of a group) on, 64-row pair tiles from 768 -- against 32-row tiles everywhere: every row is the
    same MFMA sequence, bit-identical"""
    from tests.test_gpu_round4 import _batch
    from tsdiff_amd import engine, synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(graphs, 1000, dev)
    args = (g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
    out = {}
    for wide in (True, False):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
        monkeypatch.setattr(engine.OPTIONS, "wide_filter_tiles", wide)
        ens = EnsembleSampler([make_model(cfg, 30 + k, dev) for k in range(M)])
        with torch.no_grad():
            ens(*args)
        db = ens._bound_batch(*args[:3], *args[4:7])
        assert not db.per_block
        out[wide] = db.edge_inv_u.clone()
    assert torch.equal(out[True], out[False])


def test_stand_alone_pair_launch_with_64_row_tiles_gives_the_same_bits(dev, monkeypatch):
    """the pair MLP as its own launch (behind the fused per-unit encoder) takes 64-row tiles from 4096 32-row tiles on:
    80 graphs of 62 atoms = 151 280 out pairs = 4728 tiles; OPTIONS.wide_filter_tiles = False keeps 32 rows -- bit-identical"""
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 80
    b = synth.dense_stress_batch(G, n=62, seed=4)
    g = to_dev({**{k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}, "num_graphs": G}, dev)
    out = {}
    for wide in (True, False):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", "force")
        monkeypatch.setattr(engine.OPTIONS, "wide_filter_tiles", wide)
        inv, ei, _ = run_forward(make_model(cfg, 5, dev), g, dev)
        out[wide] = inv.clone()
    assert out[True].shape[0] >= 2 * 4096 * 32 - 64
    assert torch.isfinite(out[True]).all() and torch.equal(out[True], out[False])


def test_pair_role_of_the_last_block_launch_with_64_row_tiles_gives_the_same_bits(dev, monkeypatch):
    """1300 graphs, one checkpoint (1300 node tiles: block launches, the pair MLP inside the last one): > 4096 pair tiles of
    32 rows -> 64-row tiles; OPTIONS.wide_filter_tiles = False keeps every tile at 32 rows -- bit-identical"""
    from tests.test_gpu_round4 import _batch
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(1300, 77, dev)
    out = {}
    for wide in (True, False):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
        monkeypatch.setattr(engine.OPTIONS, "wide_filter_tiles", wide)
        inv, _, _ = run_forward(make_model(cfg, 6, dev), g, dev)
        out[wide] = inv.clone()
    assert out[True].shape[0] // 2 >= 4096 * 32, out[True].shape   # (undirected out pairs: at least 4096 32-row tiles)
    assert torch.isfinite(out[True]).all() and torch.equal(out[True], out[False])
