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
    N = g["pos"].shape[0]
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
