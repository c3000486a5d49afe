"""Round-6 GPU tests (through the C ABI, `-m gpu`): the waiting pair role behind spread node tiles, the
piecewise split-f16 entry with plane rows, parity on weights the package trained itself, the per-graph aggregation."""
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
from tests.test_gpu_round4 import _db  # noqa: E402
from tests.util import assert_close  # noqa: E402

RTOL = 1e-5  # north_star: eps within 1e-5 rel-fp32 of the reference

pytestmark = pytest.mark.gpu


def _sparse_big_batch(graphs, n, box, seed, dev):
    from tsdiff_amd import synth
    b = synth.dense_stress_batch(graphs, n=n, seed=seed, box=box)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    return b, t, to_dev({**t, "num_graphs": graphs}, dev)


@pytest.mark.parametrize("graphs,n,box", [(600, 44, 16.0), (450, 44, 19.0)])
def test_pair_role_behind_many_node_tiles_with_half_the_pairs_cut_off(graphs, n, box, dev, monkeypatch):
    """>= 1024 node tiles (1650 / 1238) with the pair MLP inside the last block launch and a COMPACTED
    out list (about half / a third of the pairs inside the 10 A cutoff): the node tiles must all be dispatched ahead of the
    waiting pair tiles (node_stride 1 whenever the pair role rides along) -- no TSD_STATUS_INTERNAL, and the same bits as
    the form without any in-kernel wait (tsd_batch.reserved bit 0: stand-alone pair launch)"""
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _sparse_big_batch(graphs, n, box, 7, dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
    monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
    model = make_model(cfg, 3, dev)
    inv, ei, _ = run_forward(model, g, dev)
    db = _db(model)
    assert not db.per_block, "a bounded in-kernel wait gave up (TSD_STATUS_INTERNAL) in the default form"
    assert db.gemm_mode() == "h2"
    frac = ei.shape[1] / float(graphs * n * (n - 1))
    assert 0.2 < frac < 0.8, f"the batch does not exercise a compacted out list ({frac:.2f} of the pairs kept)"
    monkeypatch.setattr(engine.OPTIONS, "one_launch", False)
    ref_model = make_model(cfg, 3, dev)
    ref, ref_ei, _ = run_forward(ref_model, g, dev)
    assert _db(ref_model).reserved_flags() & 1  # (the form without in-kernel waits)
    assert torch.equal(ei, ref_ei)
    assert torch.equal(inv, ref)
    assert bool(torch.isfinite(inv).all())


def test_internal_status_fallback_form_has_no_waiting_pair_role(dev, monkeypatch):
    """the form the host falls back to after TSD_STATUS_INTERNAL (tsd_batch.reserved bit 0) must not
    contain the waiting pair role either (api.hip forward_impl: bit 0 selects the stand-alone pair launch).  The fault
    injection bit only reaches the one-launch kernel, so the fallback form is selected directly -- what status_fallback does --
    on a batch past the one-launch size: it carries reserved bit 0 and gives the default form's result bit for bit"""
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b, t, g = _sparse_big_batch(120, 44, 14.0, 3, dev)
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")
    monkeypatch.setattr(engine.OPTIONS, "fused_encoder", False)
    model = make_model(cfg, 3, dev)
    inv, ei, _ = run_forward(model, g, dev)
    db = _db(model)
    assert not db.per_block
    db.per_block = True  # what status_fallback does after TSD_STATUS_INTERNAL
    db.drop_plans()
    inv2, ei2, _ = run_forward(model, g, dev)
    assert torch.equal(inv2, inv) and torch.equal(ei2, ei)
    assert db.reserved_flags() & 1


def test_attr_planes_and_the_piecewise_split_f16_block_vs_the_fp32_entry(dev):
    """tsd_attr_planes + tsd_interaction_block16 (plane rows in, 0.6) against tsd_interaction_block on the
    same fp32 rows: filters of a layer and one node chain within 1e-5; tsd_attr_planes raises TSD_STATUS_RANGE for a value
    beyond the f16 range and for a run of channels that is tiny throughout, and leaves the word alone otherwise"""
    from tsdiff_amd import _lib, synth
    lib = _lib.load()
    cfg = synth.DEFAULT_MODEL_CONFIG
    H, L = cfg["hidden_dim"], cfg["encoder"]["num_convs"]
    b = synth.wb97xd3_like_batch(40, seed=4)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    g = to_dev({**t, "num_graphs": 40}, dev)
    model = make_model(cfg, 3, dev)
    run_forward(model, g, dev)
    db = _db(model)
    N, PU = db.N, db.P // 2
    gen = torch.Generator(device=dev).manual_seed(5)
    ea = torch.randn(PU, H, device=dev, generator=gen) * 0.5
    ea16 = torch.empty_like(ea)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    _lib.check(lib.tsd_attr_planes(H, PU, _lib.ptr(ea), _lib.ptr(ea16), _lib.ptr(status), _lib.stream_ptr()))
    assert int(status[0]) == 0
    # the planes reproduce the rows to 22 bits
    pl = ea16.view(torch.float16).view(PU, 2, H).float()
    back = pl[:, 0] + pl[:, 1] / 2048.0
    assert float((back - ea).abs().max()) <= 2.0 ** -21 * float(ea.abs().max())
    # (a) the filters of block 0 from plane rows.  The split-f16 filter role takes s1 and the FOLDED nn.0 (edge_cat.2
    # multiplied into it, common.hpp FOLDED WEIGHTS): reference in fp64 from the state dict -- W = (ssp(s1 nn0f^T + b) nn2^T + b2) C
    sd = {k: v.detach().double() for k, v in model.state_dict().items()}
    pre = "encoder.interactions.0.conv."
    nn0f_w = sd[pre + "nn.0.weight"] @ sd["edge_cat.2.weight"]
    nn0f_b = sd[pre + "nn.0.weight"] @ sd["edge_cat.2.bias"] + sd[pre + "nn.0.bias"]
    Eu = db.enc_u.num_edges()
    s1 = ea[:Eu].double()
    y1 = torch.nn.functional.softplus(s1 @ nn0f_w.T + nn0f_b) - np.log(2.0)
    Cw = (db.enc_u.dist[:Eu].double() <= cfg["encoder"]["cutoff"]).double().unsqueeze(-1)
    ref_w = (y1 @ sd[pre + "nn.2.weight"].T + sd[pre + "nn.2.bias"]) * Cw
    x = torch.randn(N, H, device=dev, generator=gen) * 0.3
    h0 = torch.randn(N, H, device=dev, generator=gen) * 0.3
    wf = torch.zeros(2, PU, H, device=dev)

    def blk(h2, layer, fl, xi, hbuf, xo_):
        args = (C.byref(db.cfg), _lib.ptr(db.weights16[0] if h2 else db.weights[0]), layer, N, db.enc.struct(),
                _lib.ptr(wf[layer % 2]) if layer >= 0 else None, _lib.ptr(xi), _lib.ptr(hbuf), _lib.ptr(xo_), fl, PU,
                db.enc_u.struct(), _lib.ptr(ea16 if h2 else ea), _lib.ptr(wf[fl % 2]) if fl >= 0 else None)
        if h2:
            _lib.check(lib.tsd_interaction_block16(*args, _lib.ptr(status), _lib.stream_ptr()))
        else:
            _lib.check(lib.tsd_interaction_block(*args, _lib.stream_ptr()))
    xo = torch.empty(N, H, device=dev)
    blk(True, -2, 0, x, h0.clone(), xo)
    torch.cuda.synchronize()
    assert_close(wf[0, :Eu].double().cpu().numpy(), ref_w.cpu().numpy(), RTOL, "filters of block 0 from plane rows")
    # (b) the node chain of block 0 on those filters in both arithmetics (no filter role: fl = -1)
    out = {}
    for h2 in (False, True):
        hbuf, xo = h0.clone(), torch.empty(N, H, device=dev)
        blk(h2, 0, -1, x, hbuf, xo)
        torch.cuda.synchronize()
        out[h2] = (hbuf.clone(), xo.clone())
    assert int(status[0]) == 0
    for a_, r_, what in zip(out[True], out[False], ("h", "x1")):
        assert_close(a_.cpu().numpy(), r_.cpu().numpy(), RTOL, what)
    # range reports of the producer
    big = ea.clone()
    big[3, 7] = 1.0e6
    _lib.check(lib.tsd_attr_planes(H, PU, _lib.ptr(big), _lib.ptr(ea16), _lib.ptr(status), _lib.stream_ptr()))
    assert int(status[0]) & _lib.STATUS_RANGE
    status.zero_()
    tiny = ea.clone()
    tiny[5, 16:24] = 1.0e-6  # one conversion site (8 consecutive channels of a row) below 2^-12 throughout
    _lib.check(lib.tsd_attr_planes(H, PU, _lib.ptr(tiny), _lib.ptr(ea16), _lib.ptr(status), _lib.stream_ptr()))
    assert int(status[0]) & _lib.STATUS_RANGE
    status.zero_()
    _lib.check(lib.tsd_attr_planes(H, PU, _lib.ptr(ea), _lib.ptr(ea16), None, _lib.stream_ptr()))  # NULL word: no report
    assert int(status[0]) == 0


def _ragged_csr(sizes, keep, rng):
    """CSR of a batch of graphs (contiguous node ranges): every node lists the other nodes of its graph, each kept with
    probability `keep`, ascending -- rows of 0 .. n - 1 edges, empty rows included"""
    row_ptr, dst = [0], []
    base = 0
    for n in sizes:
        for i in range(n):
            others = np.concatenate([np.arange(0, i), np.arange(i + 1, n)]) + base
            sel = others[rng.random(n - 1) < keep] if n > 1 else others
            dst.append(sel)
            row_ptr.append(row_ptr[-1] + len(sel))
        base += n
    return np.asarray(row_ptr, np.int32), (np.concatenate(dst) if dst else np.zeros(0)).astype(np.int32), base


def _sequential_rows(row_ptr, dst, widx, W, x1):
    """out[i] = sum over the row's edges IN LIST ORDER of round(x1[dst] * W[widx]), from 0: vectorised over rows, sequential
    along a row (what a sequential fp32 scatter_add does; reference schnet.py:100-107)"""
    N = row_ptr.numel() - 1
    deg = (row_ptr[1:] - row_ptr[:-1]).long()
    out = torch.zeros(N, W.shape[1], device=W.device)
    for k in range(int(deg.max()) if N else 0):
        rows = (deg > k).nonzero().view(-1)
        e = row_ptr[rows].long() + k
        out[rows] = out[rows] + x1[dst[e].long()] * W[widx[e].long()]
    return out


@pytest.mark.parametrize("H", [256, 64])
def test_windowed_aggregation_is_bitwise_the_sequential_scatter_on_ragged_graphs(H, dev):
    """cfconv_aggregate_win_kernel (round 6: a workgroup owns 32 destination rows and stages the x1 window of its edges in
    LDS once; launches of >= 16384 rows) on a batch that exercises every branch: graphs of 1 .. 130 atoms (windows that fit,
    windows that straddle graphs, graphs of more than 64 atoms -> the global-gather fallback), empty rows, rows that end
    inside / at the end of a 64-edge index chunk, a last partial workgroup, with and without the directed -> undirected
    filter map -- bit-identical to the sequential scatter in list order, and to the one-wave-per-row kernel on a slice
    small enough to take it"""
    from tsdiff_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    sizes = []
    while sum(sizes) < 17000:
        sizes.append(int(rng.choice([1, 2, 3, 7, 16, 23, 33, 40, 64, 64, 64, 65, 90, 130])))
    rp, ds, N = _ragged_csr(sizes, 0.8, rng)
    assert N >= 16384 and N % 32 != 0
    E = int(rp[-1])
    row_ptr, dst = torch.from_numpy(rp).to(dev), torch.from_numpy(ds).to(dev)
    gen = torch.Generator(device=dev).manual_seed(3)
    W = torch.randn(E, H, device=dev, generator=gen)
    x1 = torch.randn(N, H, device=dev, generator=gen)
    ident = torch.arange(E, dtype=torch.int32, device=dev)
    umap = torch.randperm(E, device=dev, generator=gen).to(torch.int32)
    for um in (None, umap):
        out = torch.full((N, H), float("nan"), device=dev)
        _lib.check(lib.tsd_cfconv_aggregate(H, N, _lib.ptr(row_ptr), _lib.ptr(dst), _lib.ptr(um) if um is not None else None,
                                            _lib.ptr(W), _lib.ptr(x1), _lib.ptr(out), _lib.stream_ptr()))
        ref = _sequential_rows(row_ptr, dst, ident if um is None else um, W, x1)
        assert torch.equal(out, ref), f"windowed aggregation differs from the sequential order (umap: {um is not None})"
    # the first 4000 rows alone (one wave per row: below the windowed form's launch size) give the same bits
    n0 = 4000
    out0 = torch.full((n0, H), float("nan"), device=dev)
    _lib.check(lib.tsd_cfconv_aggregate(H, n0, _lib.ptr(row_ptr), _lib.ptr(dst), None, _lib.ptr(W), _lib.ptr(x1),
                                        _lib.ptr(out0), _lib.stream_ptr()))
    # (rows of a graph cut by the slice still name nodes beyond it: x1 has them)
    assert torch.equal(out0, _sequential_rows(row_ptr[: n0 + 1], dst, ident, W, x1))


def test_split_f16_forward_on_weights_the_package_trained_itself(dev, monkeypatch):
    """every accuracy statement about the default (split-f16) arithmetic so far was made on closed-form
    synthetic checkpoints and perturbations of them.  Here the full H = 256 model is TRAINED -- 300 optimizer steps of the
    package's own default training step (tests/tools/split_f16_sweep.py trained_state_dict; reference train.py:124-152) -- and
    the trained weights go through the split-f16 forward, the fp32-MFMA forward and an fp64 evaluation of the pinned oracle at
    batch 100: e_h2 <= max(1.5 e_f32, 1e-6) on the tensor scale, the per-element relative error of every entry above 1 % of
    the tensor scale <= 1e-4 (both arithmetics), no range trip in training or inference, and the 1e-5 north-star bound
    against the fp32 oracle"""
    from oracle import tsdiff_oracle as O
    from tests.tools.split_f16_sweep import rel_elementwise, run_case, trained_state_dict
    from tsdiff_amd import engine, synth
    from tsdiff_amd.options import OPTIONS
    monkeypatch.setattr(engine.OPTIONS, "gemm", "h2")       # (the defaults, whatever TSDIFF_* the environment carries)
    monkeypatch.setattr(OPTIONS, "train_gemm", "h2")
    cfg = synth.DEFAULT_MODEL_CONFIG
    sd, trips, losses = trained_state_dict(cfg, 300, dev)
    assert trips == 0, f"{trips} split-f16 range trips while training"
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.8 * losses[0], "the model did not train"
    sd0 = synth.synth_state_dict(cfg, 0)
    moved = max(float(np.abs(sd[k] - sd0[k]).max()) for k in sd0 if sd0[k].ndim == 2)
    assert moved > 1e-3, "the weights did not move: the test proves nothing"
    r = run_case(cfg, sd, 100, 1000, 0.7, 9.0, dev)
    print(f"trained weights: e_f32 {r['e_f32']:.2e} e_h2 {r['e_h2']:.2e}; per-element f32 {r['elem_f32']:.2e} h2 {r['elem_h2']:.2e}; "
          f"small 1 %: f32 {r['small_f32']:.2e} h2 {r['small_h2']:.2e}; loss {losses[0]:.1f} -> {np.mean(losses[-8:]):.1f}")
    assert r["finite"] and not r["fallback"], "the split-f16 forward left the f16 range on trained weights"
    assert r["e_h2"] <= max(1.5 * r["e_f32"], 1e-6), f"split-f16 {r['e_h2']:.2e} vs fp32-MFMA {r['e_f32']:.2e} from fp64"
    assert r["elem_h2"] <= 1e-4 and r["elem_f32"] <= 1e-4
    assert r["small_h2"] <= max(1.5 * r["small_f32"], 1e-6)
    # north_star's bound against the fp32 reference arithmetic (the oracle in fp32) on the same trained weights
    b = synth.wb97xd3_like_batch(100, seed=1000)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    o32, o_ei, _ = O.forward(O.to_torch_state(sd), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"], t["bond_index"],
                             t["bond_type"], b["num_nodes_per_graph"])
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.utils import AttrDict
    model = get_model(AttrDict(cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(dev)
    inv, ei, _ = run_forward(model, to_dev({**t, "num_graphs": 100}, dev), dev)
    assert torch.equal(ei.cpu(), o_ei)
    assert_close(inv.cpu().numpy(), o32.numpy(), RTOL, "edge_inv on trained weights vs the fp32 oracle")
    assert _db(model).gemm_mode() == "h2"


def test_batch_prefetched_with_its_positions_trains_bit_identically(dev):
    """prefetch_batch(pos=...) (round 6): the next step's random draws, forward diffusion and perturbed-geometry edge lists
    are built ahead on the side stream and the training step's forward starts without its host wait for the edge counts
    (tsd_batch.reserved bit 7).  The draws are the same torch calls in the same order, so a loop that prefetches batch
    k + 1 behind step k is BIT-IDENTICAL -- loss and every parameter after four optimizer steps -- to the loop that does not
    prefetch at all and to the one that prefetches the topology only; a get_loss with another `pos` tensor ignores the stash"""
    from types import SimpleNamespace
    from tsdiff_amd import optim, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 24
    batches = [to_dev({k2: torch.from_numpy(v) for k2, v in synth.wb97xd3_like_batch(G, seed=950 + k).items()
                       if isinstance(v, np.ndarray)}, dev) for k in range(3)]

    def topo(g):
        return (g["atom_type"], g["r_feat"], g["p_feat"], g["bond_index"], g["bond_type"], g["batch"], g["num_nodes_per_graph"])

    def run(mode):
        model = make_model(cfg, 0, dev)
        model.train()
        opt = optim.get_optimizer(SimpleNamespace(type="adam", lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
        torch.manual_seed(123)
        torch.cuda.manual_seed_all(123)
        losses, used = [], 0
        for k in range(4):
            g = batches[k % 3]
            opt.zero_grad()
            hit = model._batches and getattr(model._batches[0][2], "train_stash", None) is not None
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                  g["batch"], g["num_nodes_per_graph"], G)
            used += int(bool(hit) and model._batches[0][2].train_stash is None)  # (consumed by the fused step's forward)
            loss.mean().backward()
            optim.clip_grad_norm_(model.parameters(), 3000.0)
            opt.step()
            losses.append(loss.detach().clone())
            model._batches.clear()
            nxt = batches[(k + 1) % 3]
            if mode == "topology":
                model.prefetch_batch(*topo(nxt))
            elif mode == "pos":
                db = model.prefetch_batch(*topo(nxt), pos=nxt["pos"], num_graphs=G)
                assert db.train_stash is not None
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        return losses, flat, used
    ref_l, ref_p, _ = run("none")
    for mode in ("topology", "pos"):
        l, p, used = run(mode)
        for a, b in zip(l, ref_l):
            assert torch.equal(a, b), mode
        assert torch.equal(p, ref_p), mode
        assert used == (3 if mode == "pos" else 0), f"{mode}: the stash was consumed {used} times"
    # another positions tensor than the one the batch was prefetched with: the stash is ignored (and dropped), the result is
    # what a fresh get_loss gives
    model = make_model(cfg, 0, dev)
    model.train()
    g = batches[0]
    model.prefetch_batch(*topo(g), pos=g["pos"], num_graphs=G)
    other = (g["pos"] * 1.01).contiguous()
    torch.manual_seed(5); torch.cuda.manual_seed_all(5)
    la = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], other, g["bond_index"], g["bond_type"], g["batch"],
                        g["num_nodes_per_graph"], G)
    model._batches.clear()
    torch.manual_seed(5); torch.cuda.manual_seed_all(5)
    lb = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], other, g["bond_index"], g["bond_type"], g["batch"],
                        g["num_nodes_per_graph"], G)
    assert torch.equal(la, lb)


def test_flat_gradient_form_of_the_training_step_is_bit_identical_and_falls_back(dev, monkeypatch):
    """The flat form of the fused step's autograd node (round 6; tsdiff_amd/train_ops.py fused_train_loss): with parameters
    that are views of one flat buffer and no .grad, autograd sees ONE leaf and the backward hands cached views of a
    persistent flat gradient to the Parameters.  Three optimizer steps give bit-identical losses, gradients and parameters
    to the per-parameter form (OPTIONS.train_flat_grad = False); an existing .grad (zero_grad(set_to_none=False),
    accumulation) selects the per-parameter form with torch's accumulate semantics; a re-homed Parameter is noticed"""
    from types import SimpleNamespace
    from tsdiff_amd import optim, synth
    from tsdiff_amd.options import OPTIONS
    cfg = synth.DEFAULT_MODEL_CONFIG
    G = 16
    batches = [to_dev({k2: torch.from_numpy(v) for k2, v in synth.wb97xd3_like_batch(G, seed=970 + k).items()
                       if isinstance(v, np.ndarray)}, dev) for k in range(2)]

    def run(flat_form):
        monkeypatch.setattr(OPTIONS, "train_flat_grad", flat_form)
        model = make_model(cfg, 0, dev)
        model.train()
        opt = optim.get_optimizer(SimpleNamespace(type="adam", lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
        torch.manual_seed(77)
        torch.cuda.manual_seed_all(77)
        out = []
        for k in range(3):
            g = batches[k % 2]
            opt.zero_grad()
            loss = model.get_loss(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"],
                                  g["batch"], g["num_nodes_per_graph"], G)
            assert (loss.grad_fn is not None)
            loss.mean().backward()
            grads = torch.cat([p.grad.reshape(-1) for p in model.raw_params()]).clone()
            optim.clip_grad_norm_(model.parameters(), 3000.0)
            opt.step()
            out.append((loss.detach().clone(), grads))
        torch.cuda.synchronize()
        used_flat = getattr(model, "_flat_grad_buf", None) is not None
        return out, torch.cat([p.detach().reshape(-1) for p in model.raw_params()]).clone(), used_flat, model, opt
    a, pa, fa, model, opt = run(True)
    b, pb, fb, _, _ = run(False)
    assert fa and not fb, "the flat form did not run where it should (or ran where it should not)"
    for (la, ga), (lb, gb) in zip(a, b):
        assert torch.equal(la, lb) and torch.equal(ga, gb)
    assert torch.equal(pa, pb)
    # an existing .grad: the per-parameter form, torch's accumulation (g + g)
    monkeypatch.setattr(OPTIONS, "train_flat_grad", True)
    g = batches[0]
    args = (g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"],
            g["num_nodes_per_graph"], G)
    kw = dict(_time_step=torch.arange(G, device=dev) * 37 % 5000, _pos_noise=torch.randn(g["pos"].shape, device=dev))
    opt.zero_grad()
    model.get_loss(*args, **kw).mean().backward()
    g1 = [p.grad.clone() for p in model.raw_params()]
    model.get_loss(*args, **kw).mean().backward()   # .grad exists: accumulate
    for p, x in zip(model.raw_params(), g1):
        assert torch.allclose(p.grad, 2.0 * x, rtol=1e-6, atol=0.0)
    # a frozen parameter, or a tensor hook on one: the per-parameter form (a frozen parameter gets no .grad, the hook fires)
    opt.zero_grad()
    frozen = model.raw_params()[5]
    frozen.requires_grad_(False)
    fired = []
    hk = model.raw_params()[7].register_hook(lambda gr: fired.append(1) or gr)
    model.get_loss(*args, **kw).mean().backward()
    assert frozen.grad is None and fired == [1] and model.raw_params()[6].grad is not None
    frozen.requires_grad_(True)
    hk.remove()
    # a Parameter re-homed outside the flat buffer is noticed: the step falls back to the per-parameter form (which copies
    # the parameters into a fresh flat vector) instead of training on the stale copy
    opt.zero_grad()
    p0 = model.raw_params()[3]
    p0.data = p0.data.clone() * 1.5
    l_moved = model.get_loss(*args, **kw)
    ref = make_model(cfg, 0, dev)
    ref.load_state_dict(model.state_dict())
    ref.train()
    l_ref = ref.get_loss(*args, **kw)
    assert torch.equal(l_moved, l_ref)
