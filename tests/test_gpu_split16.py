"""GPU tests of the split-f16 inference forward (csrc/split16.hpp: every tile GEMM on the f16 MFMA pipes with two-plane
operands, fp32 accumulation) and of its one-launch form (kernels_combo.hip forward_mega_kernel):
  * all three forms of the forward -- split-f16 one launch, split-f16 one launch per block, fp32 MFMA -- against the
    pinned oracle at the 1e-5 tolerance of north_star, the split forms also against an fp64 evaluation (they are closer
    to it than the fp32 chain);
  * one launch == one launch per block, bit for bit (forward, LD trajectories, graph and eager);
  * an activation beyond the f16 range raises TSD_STATUS_RANGE and the call is rerun on the fp32-MFMA kernels.
Everything goes through the C ABI (libtsdiff_hip.so)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import RTOL, _sample, _sampling_setup, make_model, run_forward, to_dev
from tests.util import assert_close, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _modes(monkeypatch):
    from tsdiff_amd import engine

    def set_mode(gemm, one_launch):
        monkeypatch.setattr(engine.OPTIONS, "gemm", gemm)
        monkeypatch.setattr(engine.OPTIONS, "one_launch", one_launch)
    return set_mode


@pytest.mark.parametrize("graphs,seed", [(1, 3), (20, 5), (100, 1000)])
def test_split_f16_forward_vs_oracle_fp64_and_fp32_mfma(graphs, seed, dev, monkeypatch):
    from oracle import tsdiff_oracle as O
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    b = synth.wb97xd3_like_batch(graphs, seed=seed)
    scale = np.repeat(np.linspace(0.7, 9.0, graphs).astype(np.float32), b["num_nodes_per_graph"])[:, None]
    b["pos"] = (b["pos"] * scale).astype(np.float32)
    t = {k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}
    sd_np = synth.synth_state_dict(cfg, 3)
    o32, o_ei, _ = O.forward(O.to_torch_state(sd_np), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"],
                             t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])
    o64 = O.forward(O.to_torch_state(sd_np, torch.float64), cfg, t["atom_type"], t["r_feat"], t["p_feat"], t["pos"].double(),
                    t["bond_index"], t["bond_type"], b["num_nodes_per_graph"])[0]
    g = to_dev({**t, "num_graphs": graphs}, dev)
    model = make_model(cfg, 3, dev)
    set_mode = _modes(monkeypatch)
    res = {}
    for name, gemm, one in (("h2_one_launch", "h2", True), ("h2_per_block", "h2", False), ("f32", "f32", True)):
        set_mode(gemm, one)
        inv, ei, _ = run_forward(model, g, dev)
        db = model._batches[0][2]
        assert db.gemm_mode() == gemm
        assert torch.equal(ei.cpu(), o_ei)
        assert_close(inv.cpu().numpy(), o32.numpy(), RTOL, f"edge_inv ({name}) vs the oracle")
        res[name] = inv.clone()
    # the one-launch form runs the same GEMMs and the same gathers as one launch per block
    assert torch.equal(res["h2_one_launch"], res["h2_per_block"])
    # against an fp64 evaluation: 22-bit operands with a separately accumulated low part are not worse than fp32 chains
    e_h2, e_f32 = rel_err(res["h2_one_launch"].cpu().numpy(), o64.numpy()), rel_err(res["f32"].cpu().numpy(), o64.numpy())
    assert e_h2 <= 2e-6 and e_h2 <= 2.0 * e_f32 + 5e-7, (e_h2, e_f32)
    # determinism (no atomics on the data path; the hand-off counters do not touch values)
    set_mode("h2", True)
    again, _, _ = run_forward(model, g, dev)
    assert torch.equal(again, res["h2_one_launch"])


@pytest.mark.parametrize("use_graph", [True, False])
def test_one_launch_sampling_equals_one_launch_per_block(use_graph, dev, monkeypatch):
    """24 LD steps (three 8-step graph launches) at 40 graphs, full model, injected noise: positions and the whole
    trajectory bit for bit; the fp32-MFMA run of the same call within the trajectory tolerance"""
    from tsdiff_amd import synth
    from tsdiff_amd.sampler import EnsembleSampler
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 0, dev)
    b = synth.wb97xd3_like_batch(40, seed=77)
    g = to_dev({k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}, dev)
    N = g["pos"].shape[0]
    gen = torch.Generator(device=dev).manual_seed(5)
    noises = torch.randn(24, N, 3, device=dev, generator=gen)
    g["pos"] = torch.randn(N, 3, device=dev, generator=gen) * 1.5
    set_mode = _modes(monkeypatch)
    out = {}
    for name, gemm, one in (("one", "h2", True), ("per_block", "h2", False), ("f32", "f32", True)):
        set_mode(gemm, one)
        pos, traj = _sample(EnsembleSampler([model]), g, 40, 24, noises=noises, use_graph=use_graph, denoise_from_time_t=24)
        out[name] = (pos.clone(), torch.stack(traj))
    assert torch.equal(out["one"][0], out["per_block"][0]) and torch.equal(out["one"][1], out["per_block"][1])
    assert_close(out["one"][1].numpy(), out["f32"][1].numpy(), 5e-6, "split-f16 trajectory vs the fp32-MFMA trajectory")
    # a second call on the cached plan (epochs of the hand-off words continue from a fresh zeroing): same result
    set_mode("h2", True)
    pos2, _ = _sample(EnsembleSampler([model]), g, 40, 24, noises=noises, use_graph=use_graph, denoise_from_time_t=24)
    assert torch.equal(pos2, out["one"][0])


def test_small_configs_and_ensembles_take_the_right_path(dev, monkeypatch):
    """H = 64 / 2 blocks, and a 2-checkpoint ensemble (per-block launches: the one-launch form is single-checkpoint):
    split-f16 == fp32 MFMA within 1e-5, one launch == per block bitwise where it applies"""
    set_mode = _modes(monkeypatch)
    for seeds in ((4,), (4, 5)):
        ens, g, G = _sampling_setup(dev, graphs=7, seed=11, model_seeds=seeds)
        res = {}
        for name, gemm, one in (("one", "h2", True), ("per_block", "h2", False), ("f32", "f32", True)):
            set_mode(gemm, one)
            inv, ei, el = ens(g["atom_type"], g["r_feat"], g["p_feat"], g["pos"], g["bond_index"], g["bond_type"], g["batch"], None)
            res[name] = inv.clone()
        assert torch.equal(res["one"], res["per_block"])
        assert_close(res["one"].cpu().numpy(), res["f32"].cpu().numpy(), RTOL, f"edge_inv, {len(seeds)} checkpoint(s)")


@pytest.mark.parametrize("n,graphs", [(40, 5), (64, 3)])
def test_one_launch_gather_paths(n, graphs, dev, monkeypatch):
    """the node workgroups of the one-launch forward gather x from an LDS copy of their graphs' rows when those are at
    most 60 (aggregate through xl_gather), from L2 otherwise: 40-atom graphs give tiles of both kinds in one launch
    (a tile inside one graph: 40 rows; across two: 80), 64-atom graphs only the second kind.  Both are bit for bit the
    forward of one launch per block"""
    from tsdiff_amd import synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 1, dev)
    b = synth.dense_stress_batch(graphs, n=n, seed=21)
    g = to_dev({**{k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}, "num_graphs": graphs}, dev)
    set_mode = _modes(monkeypatch)
    res = {}
    for name, gemm, one in (("one", "h2", True), ("per_block", "h2", False), ("f32", "f32", True)):
        set_mode(gemm, one)
        inv, _, _ = run_forward(model, g, dev)
        res[name] = inv.clone()
    assert torch.isfinite(res["one"]).all()
    assert torch.equal(res["one"], res["per_block"])
    assert_close(res["one"].cpu().numpy(), res["f32"].cpu().numpy(), RTOL, "one-launch forward vs fp32 MFMA")


def test_wide_filter_tiles_equal_narrow_ones(dev, monkeypatch):
    """a block launch with >= 1024 filter tiles takes them as 64-row tiles (two row blocks per weight fragment): every
    row is the same MFMA sequence as in a 32-row tile, so the forward is bit for bit the one with
    tsd_batch.reserved bit 1 set; both within 1e-5 of the fp32-MFMA forward.  40 graphs of 63 atoms, all pairs edges:
    78120 undirected edges = 2442 32-row tiles per block = 1220 64-row tiles + one of 40 rows"""
    from tsdiff_amd import engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 3, dev)
    b = synth.dense_stress_batch(40, n=63, seed=9)
    g = to_dev({**{k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}, "num_graphs": 40}, dev)
    set_mode = _modes(monkeypatch)
    res = {}
    for name, gemm, wide in (("wide", "h2", True), ("narrow", "h2", False), ("f32", "f32", True)):
        set_mode(gemm, True)
        monkeypatch.setattr(engine.OPTIONS, "wide_filter_tiles", wide)
        inv, _, _ = run_forward(model, g, dev)
        res[name] = inv.clone()
    assert torch.isfinite(res["wide"]).all()
    assert torch.equal(res["wide"], res["narrow"])
    assert_close(res["wide"].cpu().numpy(), res["f32"].cpu().numpy(), RTOL, "64-row filter tiles vs fp32 MFMA")


def test_activation_beyond_the_f16_range_falls_back_to_fp32(dev, monkeypatch):
    """one channel of the edge attribute is pushed to 1e5 (> 65504): the split-f16 kernels raise TSD_STATUS_RANGE and
    the host reruns the call on the fp32-MFMA kernels -- same result as asking for fp32 in the first place"""
    from tsdiff_amd import _lib, synth
    from tsdiff_amd.epsnet import get_model
    from tsdiff_amd.sampler import EnsembleSampler
    from tsdiff_amd.utils import AttrDict
    cfg = synth.small_model_config(64, 2)

    def build():
        sd = synth.synth_state_dict(cfg, 7)
        sd["edge_cat.0.bias"] = sd["edge_cat.0.bias"].copy()
        sd["edge_cat.0.bias"][5] = 1.0e5
        m = get_model(AttrDict(cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m.to(dev)
    b = synth.wb97xd3_like_batch(5, seed=2)
    g = to_dev({**{k: torch.from_numpy(v) for k, v in b.items() if isinstance(v, np.ndarray)}, "num_graphs": 5}, dev)
    set_mode = _modes(monkeypatch)
    set_mode("f32", True)
    ref_model = build()
    ref, _, _ = run_forward(ref_model, g, dev)
    assert torch.isfinite(ref).all()
    set_mode("h2", True)
    model = build()
    inv, _, _ = run_forward(model, g, dev)
    db = model._batches[0][2]
    assert db.gemm == "f32" and db.gemm_mode() == "f32"      # the batch was switched by the range flag
    assert int(db.status[0].item()) & _lib.STATUS_RANGE == 0  # ... and the flag cleared
    assert torch.equal(inv, ref)
    # the sampling loop: the whole call is rerun
    N = g["pos"].shape[0]
    noises = torch.randn(3, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    model2 = build()
    pos, traj = _sample(EnsembleSampler([model2]), g, 5, 3, noises=noises)
    set_mode("f32", True)
    rpos, rtraj = _sample(EnsembleSampler([build()]), g, 5, 3, noises=noises)
    assert torch.equal(pos, rpos) and all(torch.equal(a, b_) for a, b_ in zip(traj, rtraj))


def test_f16_plane_weight_image_round_trips(dev):
    """tsd_pack_weights16: hi + lo * 2^-11 reproduces every packed fp32 weight to 2^-22 relative, everything that is
    not a dense matrix is copied verbatim"""
    import ctypes as C
    from tsdiff_amd import _lib, engine, synth
    cfg = synth.DEFAULT_MODEL_CONFIG
    model = make_model(cfg, 2, dev)
    packed = model.packed_weights()
    lib = _lib.load()
    p16 = torch.empty_like(packed)
    _lib.check(lib.tsd_pack_weights16(C.byref(model._cfg), _lib.ptr(packed), _lib.ptr(p16), _lib.stream_ptr()))
    H = cfg["hidden_dim"]
    # bond_emb (first 100 H floats) is copied; the first dense matrix (edge_encoder.mlp.layers.1, H x H) follows the
    # two H-vectors of layers.0
    assert torch.equal(p16[: 100 * H + 2 * H], packed[: 100 * H + 2 * H])
    o = 100 * H + 2 * H
    w = packed[o:o + H * H].view(H // 4, H, 4).permute(1, 0, 2).reshape(H, H)          # [out][k]
    img = p16[o:o + H * H].view(torch.float16).view(H // 16, 2, 2, H, 8)                  # [k/16][plane][half][out][8]
    hi = img[:, 0].permute(2, 0, 1, 3).reshape(H, H).float()
    lo = img[:, 1].permute(2, 0, 1, 3).reshape(H, H).float()
    rec = hi + lo / 2048.0
    err = (rec.double() - w.double()).abs().max() / w.abs().max()
    assert float(err) < 2.0 ** -21, float(err)
    assert float((hi - w).abs().max() / w.abs().max()) < 2.0 ** -10
