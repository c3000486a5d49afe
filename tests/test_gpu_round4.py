"""GPU tests of round 4 (everything through the C ABI):
  * a bounded in-kernel wait of the one-launch forward that gives up (TSD_STATUS_INTERNAL) no longer reaches a caller:
    forward() and dynamic_sampling rerun as one launch per block, bit for bit -- by fault injection (deterministic) and
    with a second tenant holding most of the chip's workgroup slots;
  * the fused per-unit encoder (kernels_unit.hip: CFConv filters never written to memory) against the pinned oracle,
    against the materialising per-block form, and bitwise against itself;
  * the split-f16 arithmetic over weight scales, heavy tails, hidden sizes, geometry scales and small operands.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from tests.test_gpu_parity import RTOL, _sample, make_model, run_forward, to_dev
from tests.util import assert_close, rel_err

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _batch(graphs, seed, dev, scale=2.0):
    from tsdiff_amd import synth
    b = synth.wb97xd3_like_batch(graphs, seed=seed)
    b["pos"] = (b["pos"] * scale).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    return b, t, to_dev({**t, "num_graphs": graphs}, dev)


def _db(model):
    return model._batches[0][2]


# ---------------------------------------------------------------------------------------------------------------------
# TSD_STATUS_INTERNAL: the one-launch forward's bounded waits
# ---------------------------------------------------------------------------------------------------------------------
def test_one_launch_wait_that_gives_up_reruns_per_block_bit_for_bit(dev, monkeypatch):
    """fault injection (tsd_batch.reserved bit 3: the last filter tile of the last block is never run): the node
    workgroup that reads its rows waits until the bound, the launch unwinds, the status word carries
    TSD_STATUS_INTERNAL -- and forward() returns the per-block result without raising"""
    from tsdiff_amd import _lib, engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(20, 11, dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
    monkeypatch.setattr(engine.OPTIONS, "one_launch", False)
    ref_model = make_model(cfg, 3, dev)
    ref, ref_ei, _ = run_forward(ref_model, g, dev)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    model = make_model(cfg, 3, dev)
    ok, _, _ = run_forward(model, g, dev)           # the healthy one-launch forward
    assert torch.equal(ok, ref) and not _db(model).per_block
    _db(model).test_flags = 8
    inv, ei, _ = run_forward(model, g, dev)         # one wait gives up -> rerun per block
    db = _db(model)
    assert db.per_block, "the fault was not noticed"
    assert int(db.status[0].item()) & (_lib.STATUS_INTERNAL | _lib.STATUS_RANGE) == 0
    assert torch.equal(ei, ref_ei) and torch.equal(inv, ref)
    # the sampling loop: the whole call is rerun on the per-block form with the same draws
    from tsdiff_amd.sampler import EnsembleSampler
    # (dynamic_sampling scales pos_init by sigma_T ~ 12.2: start from a compact geometry, so that every pair stays inside
    # the cutoff and the skipped tile -- the last of the undirected list -- is one that node tiles really wait for)
    N = g["pos"].shape[0]
    g = dict(g, pos=g["pos"] / 12.1685)
    noises = torch.randn(4, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    monkeypatch.setattr(engine.OPTIONS, "one_launch", False)
    rpos, rtraj = _sample(EnsembleSampler([make_model(cfg, 3, dev)]), g, 20, 4, noises=noises)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    m2 = make_model(cfg, 3, dev)
    ens = EnsembleSampler([m2])
    m2.device_batch(g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"]).test_flags = 8
    pos, traj = _sample(ens, g, 20, 4, noises=noises)
    assert _db(m2).per_block
    assert torch.equal(pos, rpos) and all(torch.equal(a, b_) for a, b_ in zip(traj, rtraj))


def _occupier():
    so = os.path.join(HERE, "helpers", "liboccupy.so")
    src = os.path.join(HERE, "helpers", "occupy.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so], check=True)
    lib = C.CDLL(so)
    lib.occupy_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    lib.occupy_launch.restype = C.c_int
    return lib


def test_one_launch_forward_beside_a_second_tenant(dev, monkeypatch):
    """a long kernel of another stream holds most of the chip's workgroup slots (80 KB of LDS per workgroup: one of
    them fills half a CU) while the one-launch forward runs: whatever the dispatcher does -- the forward waits for the
    tenant, or its node workgroups start alone, starve and give up -- the caller gets the per-block result, bit for
    bit, without an exception"""
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(100, 1000, dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
    monkeypatch.setattr(engine.OPTIONS, "one_launch", False)
    ref, ref_ei, _ = run_forward(make_model(cfg, 3, dev), g, dev)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    model = make_model(cfg, 3, dev)
    run_forward(model, g, dev)  # (topology, bound weights: only the forward itself runs beside the tenant)
    occ = _occupier()
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    n_node = (g["pos"].shape[0] + 15) // 16
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    # 2 x CUs slots of 80 KB; leave exactly the node workgroups' share free
    assert occ.occupy_launch(max(2 * cus - n_node, cus), 512, 80 * 1024, 12.0, C.c_void_p(side.cuda_stream)) == 0
    inv, ei, _ = run_forward(model, g, dev)
    torch.cuda.synchronize(dev)
    print("second tenant: per_block fallback taken =", _db(model).per_block)
    assert torch.equal(ei, ref_ei) and torch.equal(inv, ref)


# ---------------------------------------------------------------------------------------------------------------------
# the fused per-unit encoder (kernels_unit.hip): filters never written to memory
# ---------------------------------------------------------------------------------------------------------------------
def _forms(monkeypatch):
    from tsdiff_amd import engine

    def set_form(form):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", form == "one_launch")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", "force" if form.startswith("fused") else False)
    return set_form


@pytest.mark.parametrize("graphs,seed", [(1, 3), (7, 4), (40, 5), (100, 1000)])
def test_fused_encoder_equals_materialised_forms_and_oracle(graphs, seed, dev, monkeypatch):
    """one workgroup per unit computes its filter tiles and consumes them in LDS / registers: edge_inv against the pinned
    oracle (1e-5), BIT-IDENTICAL to the launch-per-block and one-launch split-f16 forwards (same MFMA sequences, messages
    added in the directed CSR order), and bitwise reproducible"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _batch(graphs, seed, dev, scale=1.0)
    scale = np.repeat(np.linspace(0.7, 9.0, graphs).astype(np.float32), b["num_nodes_per_graph"])[:, None]
    t["pos"] = torch.from_numpy((b["pos"] * scale).astype(np.float32))
    g["pos"] = t["pos"].to(dev)
    sd_np = synth.synth_state_dict(cfg, 3)
    o32, o_ei, _ = O.forward(O.to_torch_state(sd_np), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"],
                             t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    set_form = _forms(monkeypatch)
    res = {}
    for form in ("per_block", "one_launch", "fused"):
        set_form(form)
        model = make_model(cfg, 3, dev)
        inv, ei, _ = run_forward(model, g, dev)
        db = _db(model)
        assert db.gemm_mode() == "h2" and not db.per_block
        assert (db.unit_node is not None) and int(db.unit_node[-1]) == db.N
        assert torch.equal(ei.cpu(), o_ei)
        assert_close(inv.cpu().numpy(), o32.numpy(), RTOL, f"edge_inv ({form}) vs the oracle")
        res[form] = inv.clone()
        if form == "fused":
            again, _, _ = run_forward(model, g, dev)
            assert torch.equal(again, inv)
    assert torch.equal(res["fused"], res["per_block"])
    assert torch.equal(res["fused"], res["one_launch"])


def test_fused_encoder_ensemble_and_sampling_loop(dev, monkeypatch):
    """M = 3 checkpoints in one launch (grid.y) and the device-resident LD loop (graph replay) on the fused encoder:
    bit-identical to the per-block form, trajectory included"""
    from tsdiff_amd import synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.DEFAULT_MODEL_CONFIG
    _, _, g = _batch(30, 21, dev)
    g = dict(g, pos=g["pos"] / 12.1685)
    N = g["pos"].shape[0]
    noises = torch.randn(5, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(9))
    set_form = _forms(monkeypatch)
    out = {}
    for form in ("per_block", "fused"):
        set_form(form)
        models = [make_model(cfg, s, dev) for s in (1, 2, 3)]
        ens = EnsembleSampler(models)
        inv, _, _ = ens(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"] * 12.1685 * 0.3, g["bond_index"], g["bond_type"],
                        g["batch"], torch.zeros(30, dtype=torch.long, device=dev))
        pos, traj = _sample(ens, g, 30, 5, noises=noises)
        out[form] = (inv.clone(), pos.clone(), [x.clone() for x in traj])
    assert torch.equal(out["fused"][0], out["per_block"][0])
    assert torch.equal(out["fused"][1], out["per_block"][1])
    assert all(torch.equal(a, b_) for a, b_ in zip(out["fused"][2], out["per_block"][2]))


def test_fused_encoder_config_c5_shape_and_partial_runs(dev, monkeypatch):
    """BASELINE configs[4] shape (64-atom graphs, complete pair sets: 2016 pairs = 31.5 tiles per unit) on the fused
    encoder: against the oracle and bit-identical to the per-block form; a run split in two launches (blocks [0,3) and
    [3,L)) through tsd_forward_encoder lands on the same h"""
    from oracle import tsdiff_oracle as O
    from tests.test_gpu_parity import _dense_batch
    from tsdiff_amd import _lib, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _dense_batch(8, 5, dev)
    o_inv, o_ei, _ = O.forward(O.to_torch_state(synth.synth_state_dict(cfg, 1)), cfg, t["atom_type"], t["r_feat"],
                               t["p_feat"], t["pos"], t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    set_form = _forms(monkeypatch)
    res = {}
    for form in ("per_block", "fused"):
        set_form(form)
        model = make_model(cfg, 1, dev)
        inv, ei, _ = run_forward(model, g, dev)
        assert torch.equal(ei.cpu(), o_ei)
        assert_close(inv.cpu().numpy(), o_inv.numpy(), RTOL, f"edge_inv (C5 shape, {form})")
        res[form] = inv.clone()
        assert _db(model).units_single_graph
    assert torch.equal(res["fused"], res["per_block"])
    # the encoder alone, whole and in two parts, on the state the forward left in the workspace
    lib = _lib.load()
    db = _db(model)
    N, H, L = db.N, 256, cfg["encoder"]["num_convs"]
    bs = db.struct()
    ws = db.workspace
    lay = (C.c_size_t * 8)()
    _lib.check(lib.tsd_forward_workspace_layout(C.byref(db.cfg), db.N, db.P, 1, lay))
    h_off = int(lay[2])
    def h_now():
        return ws[h_off:h_off + N * H].clone()
    _lib.check(lib.tsd_forward_encoder(C.byref(db.cfg), C.byref(bs), 0, L, _lib.stream_ptr()))
    whole = h_now()
    _lib.check(lib.tsd_forward_encoder(C.byref(db.cfg), C.byref(bs), 0, 3, _lib.stream_ptr()))
    _lib.check(lib.tsd_forward_encoder(C.byref(db.cfg), C.byref(bs), 3, L, _lib.stream_ptr()))
    assert torch.equal(h_now(), whole) and bool(torch.isfinite(whole).all()) and float(whole.abs().max()) > 0


@pytest.mark.parametrize("n,graphs,cut", [(37, 5, 10.0), (50, 3, 10.0), (64, 2, 4.0), (33, 4, 3.0)])
def test_fused_encoder_block_tiles_on_ragged_and_sparse_graphs(n, graphs, cut, dev, monkeypatch):
    """the 8 x 8 atom-block tiles of the fused encoder on graphs whose atom count is no multiple of 8 and,
    with a short radius cutoff, on graphs where most atom pairs are NOT edges (zero rows inside the tiles): bit-identical
    to the launch-per-block form, and against the oracle"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    cfg = dict(synth.DEFAULT_MODEL_CONFIG)
    cfg["edge_cutoff"] = cut
    cfg["encoder"] = dict(cfg["encoder"], cutoff=cut)
    b = synth.dense_stress_batch(graphs, n=n, seed=3)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    g = to_dev({**t, "num_graphs": graphs}, dev)
    o_inv, o_ei, _ = O.forward(O.to_torch_state(synth.synth_state_dict(cfg, 2)), cfg, t["atom_type"], t["r_feat"],
                               t["p_feat"], t["pos"], t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    set_form = _forms(monkeypatch)
    res = {}
    for form in ("per_block", "fused"):
        set_form(form)
        model = make_model(cfg, 2, dev)
        inv, ei, _ = run_forward(model, g, dev)
        assert torch.equal(ei.cpu(), o_ei)
        assert_close(inv.cpu().numpy(), o_inv.numpy(), RTOL, f"edge_inv (n = {n}, cutoff {cut}, {form})")
        res[form] = inv.clone()
    if cut < 10.0:
        assert ei.shape[1] < graphs * n * (n - 1), "the short cutoff was meant to drop pairs"
    assert torch.equal(res["fused"], res["per_block"])


# ---------------------------------------------------------------------------------------------------------------------
# the split-f16 arithmetic away from torch-default magnitudes
# ---------------------------------------------------------------------------------------------------------------------
SWEEP = [("default", 256, 7, 1.0, False, (0.7, 9.0)), ("weights x 0.1", 256, 7, 0.1, False, (0.7, 9.0)),
         ("weights x 2", 256, 7, 2.0, False, (0.7, 9.0)), ("heavy tail", 256, 7, 1.0, True, (0.7, 9.0)),
         ("compact", 256, 7, 1.0, False, (0.3, 1.0)), ("stretched", 256, 7, 1.0, False, (4.0, 12.0)),
         ("hidden 128", 128, 4, 1.0, False, (0.7, 9.0)), ("hidden 64 x 0.1", 64, 3, 0.1, False, (0.7, 9.0))]


@pytest.mark.parametrize("case", SWEEP, ids=[c[0] for c in SWEEP])
def test_split_f16_sweep_vs_fp64(case, dev):
    """weight scale, weight distribution, hidden size, geometry scale: per tensor the split-f16 forward is as close to an
    fp64 evaluation as the fp32-MFMA forward (e_h2 <= max(1.5 e_f32, 1e-6)), also on the 1 % smallest entries (absolute
    error over the tensor scale); a case that leaves the f16 range is rerun in fp32 by itself (then both are that path)"""
    from tests.tools.split_f16_sweep import config_for, run_case, scaled_state_dict
    name, H, L, ws, heavy, (lo, hi) = case
    cfg = config_for(H, L)
    r = run_case(cfg, scaled_state_dict(cfg, 3, ws, heavy), 40, 1000, lo, hi, dev)
    assert r["finite"], "the fp64 reference itself is not finite: not a usable case"
    print(name, r)
    assert r["e_h2"] <= max(1.5 * r["e_f32"], 1e-6), r
    assert r["small_h2"] <= max(2.0 * r["small_f32"], 1e-6), r
    assert r["preflight"]["beyond_f16_range"] == 0


@pytest.mark.parametrize("role,param,index", [("embedding", "edge_cat.0.bias", 5), ("filter", "encoder.interactions.1.conv.nn.0.bias", 3),
                                               ("node", "encoder.interactions.0.conv.lin2.bias", 7),
                                               ("pair", "grad_dist_mlp.layers.0.bias", 2)])
def test_range_flag_trips_in_every_role(role, param, index, dev, monkeypatch):
    """an activation beyond 65504 produced in the embedding tile, a filter tile, the node chain or the pair MLP: each role
    reports TSD_STATUS_RANGE, the call is rerun on the fp32-MFMA kernels and equals an fp32 call; for the forms of the
    forward that have these roles as separate kernels / as roles of one launch / inside the fused encoder"""
    from tsdiff_amd import engine, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    cfg = synth.small_model_config(256, 2)  # (two blocks: 1e5 injected into the full 7-block model overflows fp32 itself)

    def build():
        sd = synth.synth_state_dict(cfg, 7)
        sd[param] = sd[param].copy()
        sd[param][index] = 1.0e5
        m = get_model(AttrDict(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m.to(dev)
    _, _, g = _batch(6, 2, dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "f32")
    ref, _, _ = run_forward(build(), g, dev)
    assert torch.isfinite(ref).all()
    for form in ("one_launch", "per_block", "fused"):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", form == "one_launch")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", "force" if form == "fused" else False)
        model = build()
        inv, _, _ = run_forward(model, g, dev)
        db = _db(model)
        assert db.gemm == "f32", f"{role}: the range flag did not trip in the {form} form"
        assert torch.equal(inv, ref), (role, form)


def test_operands_in_the_f16_subnormal_band(dev, monkeypatch):
    """GEMM A operands between 1e-6 and 6e-5 (below the f16 normal range: the high plane is a subnormal, the value is
    carried by the scaled low plane with an ABSOLUTE precision of ~1.5e-11): the attribute rows of every edge are pushed
    there by a 2e-5 scale on edge_cat.0, compensated in edge_cat.2 (so the network's function and output scale stay put).
    Left alone the split-f16 arithmetic lands 1.4e-5 from the fp64 evaluation there (measured, round 4: outside
    north_star's 1e-5).  The conversion sites therefore watch the LOW side of the range too (split16.hpp site_close): a
    tile whose attribute rows are tiny throughout reports TSD_STATUS_RANGE and the call is rerun on the fp32-MFMA
    kernels -- in every form of the forward -- and the result is the fp32 forward's, 1e-6 from fp64."""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import engine, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    cfg = synth.DEFAULT_MODEL_CONFIG
    sc = 2.0e-5
    sd = {k: v.copy() for k, v in synth.synth_state_dict(cfg, 3).items()}
    sd["edge_cat.0.weight"] *= np.float32(sc)
    sd["edge_cat.0.bias"] *= np.float32(sc)
    sd["edge_cat.2.weight"] *= np.float32(2.0 / sc)  # (swish(x) ~ x / 2 for small x)
    b, t, g = _batch(30, 5, dev)
    o64 = O.forward(O.to_torch_state(sd, torch.float64), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].double(),
                    t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])[0].numpy()

    def build():
        m = get_model(AttrDict(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m.to(dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "f32")
    ref, _, _ = run_forward(build(), g, dev)
    e_f32 = rel_err(ref.cpu().numpy(), o64)
    assert e_f32 <= 5e-6, e_f32
    for form in ("one_launch", "per_block", "fused"):
        monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
        monkeypatch.setattr(engine.OPTIONS, "one_launch", form == "one_launch")
        monkeypatch.setattr(engine.OPTIONS, "fused_encoder", "force" if form == "fused" else False)
        m = build()
        inv, _, _ = run_forward(m, g, dev)
        assert _db(m).gemm == "f32", f"tiny attribute rows were not noticed in the {form} form"
        assert torch.equal(inv, ref)
    # a healthy checkpoint does not trip the low side (no false positive on the default weights)
    monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
    monkeypatch.setattr(engine.OPTIONS, "one_launch", True)
    h = make_model(cfg, 3, dev)
    run_forward(h, g, dev)
    assert _db(h).gemm is None and _db(h).gemm_mode() == "h2"


# ---------------------------------------------------------------------------------------------
# The training step on split-f16 operands (OPTIONS.train_gemm = "h2": saving forms of the block launches, backward filter
# chain and the filter MLPs' batched weight gradients with gradient operands scaled by powers of two)
# ---------------------------------------------------------------------------------------------
def _train_case(dev, graphs=12, seed=31):
    from tsdiff_amd import synth
    b = synth.wb97xd3_like_batch(graphs, seed=seed)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    t["pos"] = t["pos"] * 1.5
    gen = torch.Generator().manual_seed(3)
    ts = torch.randint(0, 5000, (graphs,), generator=gen)
    pn = torch.randn(t["pos"].shape, generator=gen)
    return to_dev(t, dev), ts.to(dev), pn.to(dev), graphs


def _train_step_grads(model, g, ts, pn, G, mode, monkeypatch, loss_scale=1.0, keep_mode=False):
    from tsdiff_amd.options import OPTIONS
    monkeypatch.setattr(OPTIONS, "train_gemm", mode)
    if not keep_mode:
        model._train_f32 = False
    model.train()
    model.zero_grad(set_to_none=True)
    loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                          g["batch"], g["num_nodes_per_graph"], G, _time_step=ts, _pos_noise=pn)
    (loss.mean() * loss_scale).backward()
    torch.cuda.synchronize()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("loss_scale", [1.0, 1e-12, 1e9])
def test_train_split_f16_matches_fp32(loss_scale, dev, monkeypatch):
    """every parameter gradient of one training step in split-f16 arithmetic against the fp32-MFMA step: max|d| <= 5e-6
    of each tensor's scale (measured 7e-7) -- also with the loss scaled by 1e-12 / 1e9, i.e. with every gradient operand
    far outside the band in which an UNSCALED f16 split keeps its bits (the dgrad tiles and the dY tensors of the weight
    gradients are scaled by exact powers of two: split16.hpp, GRADIENT operands)"""
    from tsdiff_amd import synth
    model = make_model(synth.DEFAULT_MODEL_CONFIG, 2, dev)
    g, ts, pn, G = _train_case(dev)
    lf, gf = _train_step_grads(model, g, ts, pn, G, "f32", monkeypatch, loss_scale)
    lh, gh = _train_step_grads(model, g, ts, pn, G, "h2", monkeypatch, loss_scale)
    assert not getattr(model, "_train_f32", False), "the split-f16 step fell back to fp32"
    assert float((lh - lf).abs().max()) <= 2e-6 * float(lf.abs().max())
    assert len(gf) == 7 + 9 * 7 + 6 + 4
    differs = 0
    for k, ref in gf.items():
        scale = float(ref.abs().max())
        err = float((gh[k] - ref).abs().max())
        assert np.isfinite(err) and err <= 5e-6 * max(scale, 1e-30), f"{k}: {err:.3e} vs scale {scale:.3e}"
        differs += int(err > 0)
    assert differs > 40  # (the two runs really took different kernels)


def test_train_split_f16_range_fallback(dev, monkeypatch):
    """an activation beyond the f16 range inside the split-f16 training forward: tsd_train_backward2 returns TSD_ERR_RANGE
    before launching anything, the host recomputes the forward on the fp32 kernels into the same loss tensor and runs the
    fp32 backward -- loss and gradients are then BIT-identical to a step that ran in fp32 from the start.  The fallback is
    PER STEP (round 5): the next step tries split-f16 again (and trips again on this model), the trips are counted, and
    only OPTIONS.train_fallback_latch trips in a row latch the model to fp32"""
    from tsdiff_amd import synth
    from tsdiff_amd.options import OPTIONS
    model = make_model(synth.DEFAULT_MODEL_CONFIG, 2, dev)
    with torch.no_grad():
        model.encoder.interactions[3].conv.lin2.bias.add_(3.0e5)  # x2 of block 3 ~ 3e5: ssp(x2) leaves the f16 range
    g, ts, pn, G = _train_case(dev)
    lf, gf = _train_step_grads(model, g, ts, pn, G, "f32", monkeypatch)
    assert np.isfinite(float(lf.abs().max()))
    monkeypatch.setattr(OPTIONS, "train_fallback_latch", 3)
    for trip in (1, 2):
        with pytest.warns(RuntimeWarning, match="split-f16 training"):
            lh, gh = _train_step_grads(model, g, ts, pn, G, "h2", monkeypatch)
        assert not getattr(model, "_train_f32", False)  # not sticky
        assert model._h2_range_trips == trip and model._h2_range_run == trip
        assert torch.equal(lh, lf)
        for k, ref in gf.items():
            assert torch.equal(gh[k], ref), k
    # the third trip in a row latches: warning says so, and the step after it runs in fp32 without a warning
    with pytest.warns(RuntimeWarning, match="now trains in fp32"):
        lh, gh = _train_step_grads(model, g, ts, pn, G, "h2", monkeypatch)
    assert model._train_f32 is True and torch.equal(lh, lf)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model.zero_grad(set_to_none=True)
        loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                              g["batch"], g["num_nodes_per_graph"], G, _time_step=ts, _pos_noise=pn)
        loss.mean().backward()
    assert OPTIONS.train_gemm == "h2" and torch.equal(loss.detach(), lf)
    # a step inside the range resets the run counter (another model: same code path)
    ok = make_model(synth.DEFAULT_MODEL_CONFIG, 2, dev)
    ok._h2_range_run = 2
    _train_step_grads(ok, g, ts, pn, G, "h2", monkeypatch)
    assert ok._h2_range_run == 0 and not getattr(ok, "_train_f32", False)


def test_train_split_f16_edge_cases(dev, monkeypatch):
    """the split-f16 step on degenerate batches of the production width: a one-atom graph + an edge-less graph + a bonded
    triple (partial tiles everywhere), and a batch with NO edge at all (every tile kernel and weight-gradient launch sees
    zero rows) -- finite, and equal to the fp32 step within the split-f16 tolerance"""
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 4, dev)
    F = cfg["feat_dim"]
    atom = torch.tensor([6, 1, 1, 6, 1, 8], device=dev)
    feat = torch.zeros(6, F, dtype=torch.long, device=dev)
    pos = torch.tensor([[0, 0, 0], [0, 0, 0], [50., 0, 0], [0, 0, 0], [1., 0, 0], [0, 1.2, 0]], device=dev)
    batch = torch.tensor([0, 1, 1, 2, 2, 2], device=dev)
    nn_ = torch.tensor([1, 2, 3], device=dev)
    ts = torch.tensor([100, 2500, 4000], device=dev)
    pn = torch.randn(6, 3, generator=torch.Generator().manual_seed(1)).to(dev) * 0.01
    cases = {"bonded triple": (torch.tensor([[3, 4, 3, 5], [4, 3, 5, 3]], device=dev), torch.tensor([23, 23, 22, 22], device=dev)),
             "no edges": (torch.zeros(2, 0, dtype=torch.long, device=dev), torch.zeros(0, dtype=torch.long, device=dev))}
    for name, (bi, bt) in cases.items():
        p = pos.clone()
        if name == "no edges":
            p[3:] = torch.tensor([[0., 0, 0], [60., 0, 0], [0, 70., 0]], device=dev)
        g = {"atom_type": atom, "r_feat": feat, "p_feat": feat, "pos": p, "bond_index": bi, "bond_type": bt, "batch": batch,
             "num_nodes_per_graph": nn_}
        lf, gf = _train_step_grads(model, g, ts, pn, 3, "f32", monkeypatch)
        lh, gh = _train_step_grads(model, g, ts, pn, 3, "h2", monkeypatch)
        assert not getattr(model, "_train_f32", False), name
        assert torch.isfinite(lh).all() and float((lh - lf).abs().max()) <= 2e-6 * max(float(lf.abs().max()), 1e-30), name
        for k, ref in gf.items():
            assert torch.isfinite(gh[k]).all(), (name, k)
            scale = float(ref.abs().max())
            assert float((gh[k] - ref).abs().max()) <= 5e-6 * scale + 1e-30, (name, k)


def test_train_side_lane_is_bit_identical(dev, monkeypatch):
    """the split-f16 backward with its small gradient launches on the library's side stream (default) and on the caller's
    stream: the same kernels in another schedule -- bit-identical loss and gradients, three steps in a row each"""
    from tsdiff_amd import synth
    from tsdiff_amd.options import OPTIONS
    model = make_model(synth.DEFAULT_MODEL_CONFIG, 2, dev)
    g, ts, pn, G = _train_case(dev, graphs=24, seed=9)
    res = {}
    for lane in (True, False):
        monkeypatch.setattr(OPTIONS, "train_side_lane", lane)
        for _ in range(3):
            res[lane] = _train_step_grads(model, g, ts, pn, G, "h2", monkeypatch)
    assert torch.equal(res[True][0], res[False][0])
    for k, ref in res[False][1].items():
        assert torch.equal(res[True][1][k], ref), k


def test_train_split_f16_batch200_vs_oracle(dev):
    """BASELINE configs[3] at its full size (200 graphs, ~3000 atoms, production network) in the default split-f16
    arithmetic: loss and EVERY parameter gradient, element by element, against the pinned oracle's autograd on the host
    (max|d| <= 2e-5 of each tensor's scale -- the bar of the fp32 step)"""
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    from tsdiff_amd.options import OPTIONS
    assert OPTIONS.train_gemm == "h2"
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 200
    b = synth.wb97xd3_like_batch(G, seed=2000)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    t["pos"] = t["pos"] * 1.5
    gen = torch.Generator().manual_seed(11)
    ts = torch.randint(0, 5000, (G,), generator=gen)
    pn = torch.randn(t["pos"].shape, generator=gen)
    model = make_model(cfg, 2, dev)
    model.train()
    model.zero_grad()
    g = to_dev(t, dev)
    loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
                          g["num_nodes_per_graph"], G, _time_step=ts.to(dev), _pos_noise=pn.to(dev))
    loss.mean().backward()
    assert not getattr(model, "_train_f32", False)
    osd = O.to_torch_state(synth.synth_state_dict(cfg, 2))
    for v in osd.values():
        v.requires_grad_(True)
    o_loss = O.get_loss(osd, cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"], t["bond_type"],
                        t["batch"], t["num_nodes_per_graph"].numpy(), ts, pn)
    o_loss.mean().backward()
    assert_close(loss.detach().cpu().numpy(), o_loss.detach().numpy(), 5e-5, "loss at batch 200")
    P = dict(model.named_parameters())
    n, worst = 0, 0.0
    for k, v in osd.items():
        if v.grad is None or k in ("betas", "alphas"):
            continue
        ref = v.grad.numpy()
        err = float(np.abs(P[k].grad.cpu().numpy() - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        worst = max(worst, err)
        assert err <= 2e-5, f"d loss / d {k}: {err:.3e} of the tensor's scale"
        n += 1
    assert n == 7 + 9 * 7 + 6 + 4
    print(f"worst gradient deviation from the oracle at batch 200: {worst:.2e}")
