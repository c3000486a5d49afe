"""CPU: the N>1 sharding path with world_size-2 gloo (no GPU): shard balance, order-preserving gather."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tsdiff_amd import synth
from tsdiff_amd.distributed import sample_sharded, shard_bounds


def test_shard_bounds_balanced_and_contiguous():
    rng = np.random.default_rng(0)
    n = rng.integers(8, 24, size=2400)
    for world in (1, 2, 4, 8):
        b = shard_bounds(n, world)
        assert b[0] == 0 and b[-1] == len(n) and all(b[i] <= b[i + 1] for i in range(world))
        work = n * (n - 1)
        per = [work[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(per) <= 1.02 * work.sum() / world + work.max()
    assert shard_bounds([5], 4) == [0, 0, 0, 0, 1] or shard_bounds([5], 4)[-1] == 1
    assert shard_bounds([], 2) == [0, 0, 0]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = synth.wb97xd3_like_batch(11, seed=4)
    off = np.concatenate([[0], np.cumsum(b["num_nodes_per_graph"])])
    graphs = [{"atom_type": torch.from_numpy(b["atom_type"][off[g]:off[g + 1]]),
               "pos": torch.from_numpy(b["pos"][off[g]:off[g + 1]])} for g in range(11)]

    def fake_sampler(shard, r):  # deterministic function of the graph: shows order is preserved
        return [g["pos"] * 2.0 + g["atom_type"].float().unsqueeze(-1) for g in shard]

    res = sample_sharded(graphs, fake_sampler)
    if rank == 0:
        ok = len(res) == 11 and all(
            torch.equal(res[g], graphs[g]["pos"] * 2.0 + graphs[g]["atom_type"].float().unsqueeze(-1))
            for g in range(11))
        ret["ok"] = bool(ok)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sample_sharded_gloo_world2():
    port = 29500 + (os.getpid() % 500)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret.get("ok") is True


def _dp_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsdiff_amd.distributed import dp_backward
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    x = torch.randn(23, 5)  # "nodes" of the global batch; ranks own unequal shards
    sl = slice(0, 9) if rank == 0 else slice(9, 23)
    mean = dp_backward(model, model(x[sl]) ** 2)
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # single-process reference: loss.mean().backward() over all 23 nodes
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    ref.load_state_dict(model.state_dict())
    full = ref(x) ** 2
    full.mean().backward()
    gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    ok = torch.allclose(grads, gref, rtol=1e-5, atol=1e-7) and abs(mean - float(full.mean())) < 1e-6
    ok = ok and model._last_reduce == "gather-scatter"
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_dp_backward_matches_single_process_gloo_world2():
    port = 29600 + (os.getpid() % 300)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret.get(0) is True and ret.get(1) is True


def _worker4(rank, world, port, ret):
    """world-size 4, three graphs of very different sizes: shards are uneven and at least one rank gets NO graph"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [23, 3, 9]
    rng = np.random.default_rng(7)
    graphs = [{"atom_type": torch.from_numpy(rng.integers(1, 9, size=n)), "pos": torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32))}
              for n in sizes]
    b = shard_bounds(sizes, world)
    mine = b[rank + 1] - b[rank]
    seen = []

    def fake_sampler(shard, r):
        seen.append(len(shard))
        return [g["pos"] * -1.5 + g["atom_type"].float().unsqueeze(-1) for g in shard]

    res = sample_sharded(graphs, fake_sampler)
    ok = (seen == [mine]) if mine else (seen == [])  # an empty shard never calls the sampler
    if rank == 0:
        ok = ok and len(res) == 3 and all(
            torch.equal(res[g], graphs[g]["pos"] * -1.5 + graphs[g]["atom_type"].float().unsqueeze(-1)) for g in range(3))
        ret["bounds"] = list(b)
    else:
        ok = ok and res is None
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sample_sharded_gloo_world4_uneven_and_empty_shards():
    """reference sampling.py:169-231 over 4 ranks when the graph list is shorter than the world: contiguous shards of
    sizes (1, 0, 1, 1) or similar -- the rank with the empty shard takes part in the gather, the order of the results is
    the original one"""
    port = 29900 + (os.getpid() % 90)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker4, args=(4, port, ret), nprocs=4, join=True)
        assert all(ret.get(r) is True for r in range(4)), dict(ret)
        b = ret["bounds"]
        assert b[0] == 0 and b[-1] == 3 and min(b[i + 1] - b[i] for i in range(4)) == 0  # one shard is empty


def _dp_worker4(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tsdiff_amd.distributed import dp_backward
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    x = torch.randn(23, 5)
    cuts = [0, 9, 9, 20, 23]  # rank 1 owns NO node (an empty shard of the global batch)
    sl = slice(cuts[rank], cuts[rank + 1])
    mean = dp_backward(model, model(x[sl]) ** 2)
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 1))
    ref.load_state_dict(model.state_dict())
    full = ref(x) ** 2
    full.mean().backward()
    gref = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    ret[rank] = bool(torch.allclose(grads, gref, rtol=1e-5, atol=1e-7) and abs(float(mean) - float(full.mean())) < 1e-6)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_dp_backward_gloo_world4_with_an_empty_shard():
    """train.py:140-145 over 4 ranks, one of which holds no node: the global-mean loss and the summed gradient equal the
    single-process ones on EVERY rank (the empty rank contributes zeros to both all-reduces)"""
    port = 29700 + (os.getpid() % 90)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dp_worker4, args=(4, port, ret), nprocs=4, join=True)
        assert all(ret.get(r) is True for r in range(4)), dict(ret)
