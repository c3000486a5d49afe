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
