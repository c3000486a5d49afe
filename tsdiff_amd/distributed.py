"""Multi-GPU sampling: independent reaction graphs shard across ranks, one process per GPU.

Graphs never interact on this path (radius graph per `batch`, intra-graph scatters, per-graph
centring: reference models/common.py:344, models/sampler.py:260-262), so there is NO data-path
collective: every rank holds all M checkpoints, samples its own contiguous shard, and the
positions are gathered once at the end (host side, original order) -- the `sampling.py:218-231`
unbatching step.  torch.distributed ("nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used for
that final gather only.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(num_nodes_per_graph, world_size):
    """Contiguous shards balanced by the per-graph work  n*(n-1)  (ordered pairs = edges).
    Returns `world_size + 1` graph offsets; shard r = graphs[b[r]:b[r+1]]."""
    n = np.asarray(num_nodes_per_graph, dtype=np.int64)
    work = np.maximum(n * (n - 1), 1)
    cum = np.concatenate([[0], np.cumsum(work)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        k = int(np.searchsorted(cum, target, side="left"))
        # closest prefix to the target, never moving backwards
        if k > 0 and abs(cum[k - 1] - target) <= abs(cum[min(k, len(cum) - 1)] - target):
            k -= 1
        bounds.append(max(bounds[-1], min(k, len(n))))
    bounds.append(len(n))
    return bounds


def sample_sharded(graphs, sample_fn, group=None):
    """graphs: list of per-graph dicts (same list on every rank).  `sample_fn(shard_graphs, rank)`
    returns one (n_g, 3) position tensor per graph of the shard.  Rank 0 returns the positions of
    ALL graphs in the original order; other ranks return None."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nn = [int(g["atom_type"].shape[0]) for g in graphs]
    b = shard_bounds(nn, world)
    mine = graphs[b[rank]:b[rank + 1]]
    out = sample_fn(mine, rank) if mine else []
    out = [torch.as_tensor(p).detach().cpu() for p in out]
    assert len(out) == len(mine)
    if world == 1:
        return out
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(out, gathered, dst=0, group=group)
    if rank != 0:
        return None
    res = [p for part in gathered for p in part]
    assert len(res) == len(graphs)
    return res


# -------------------------------------------------------------------------------------------------
# data-parallel training step (BASELINE config 4): graphs shard across ranks, ONE collective
# -------------------------------------------------------------------------------------------------
def _all_reduce_sum(t, group=None):
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def dp_backward(model, loss_nodes, group=None, reduce_fn=None, always_reduce=False, overlap=None):
    """Backward of the reference's `loss.mean()` (train.py:140-143) over the GLOBAL batch.

    `loss_nodes` is this rank's (N_r, 1) per-node loss.  The reference averages over all nodes of the
    batch, so every rank back-propagates  sum(loss_r) / N_global  and the parameter gradients are summed
    with one all-reduce of the flat fp32 gradient (2.77 M floats = 11 MB for the shipped config; RCCL over
    xGMI on the GPUs, gloo in the CPU tests).  Afterwards `clip_grad_norm_` and the optimizer step see
    exactly the single-process gradients on every rank.  Returns the global mean loss as a 0-dim device
    tensor (no host sync anywhere in here: the host keeps enqueueing the optimizer step and the next batch
    while the GPU is still in the backward pass; call `.item()` when a number is needed).

    `reduce_fn(tensor)` replaces the in-place sum all-reduce (tests: a reducer that emulates a second rank
    exercises the flat-gradient path of the real model on one GPU); with it the reduction runs whether or not
    a process group exists.  `always_reduce`: run the collectives on a one-rank group too (bench.py under
    `torch.distributed.run --nproc-per-node 1`: the timeline of the overlap below on a 1-GPU lease).

    OVERLAP.  The fused training step's backward (one C call) reports, through an event, the moment the gradients of
    the interaction blocks are final -- one contiguous range of the flat vector, 83 % of it; their all-reduce starts
    right there on a side stream and runs beside the rest of the backward pass (the embedding's backward chain, the
    node-embedding gradients); the two remaining ranges (head and tail of the vector) are reduced when the backward has
    finished.  Three collectives per step in a fixed order on every rank.  `overlap=False`: ONE all-reduce of the whole
    flat gradient behind the backward pass (the form of rounds 1-3; A/B: `bench.py --workload train --single-range-reduce`).
    `overlap=None` (default): OPTIONS.dp_overlap, which is OFF -- the three-range form costs ~0.14 ms of HOST time per
    step (an event, a side stream, three collective calls: profiles/r05_bench_train_torchrun_*.json, 2.28 vs 2.13 ms on
    one rank) and hides at most the blocks' 83 % of an 11-MB ring all-reduce (~0.11 of ~0.13 ms on 8 ranks over xGMI,
    DESIGN.md section 6); while the step is as long as its host side (2.0 ms of launches for 2.0 ms of kernels) that is
    a net loss on any world size.  It pays once the host runs ahead of the GPU by more than those 0.14 ms per step, or
    on fabrics where the all-reduce is several times slower than xGMI (multi-node): opt in with OPTIONS.dp_overlap /
    TSDIFF_DP_OVERLAP=1 / overlap=True.  No measured multi-rank number exists yet (no N > 1 box has run this code)."""
    if overlap is None:
        from .options import OPTIONS
        overlap = bool(OPTIONS.dp_overlap)
    distributed = reduce_fn is not None or (dist.is_initialized() and (always_reduce or dist.get_world_size(group) > 1))
    reduce = reduce_fn if reduce_fn is not None else (lambda t: _all_reduce_sum(t, group))
    if not distributed:
        # one process: the global node count is this rank's -- d(mean)/d(loss_i) = 1/N as ONE fill launch instead of the
        # sum / stack / divide graph below and its three backward launches (same value: both are the correctly rounded
        # fp32 quotient 1/N)
        n = float(loss_nodes.shape[0])
        model._dp_early_reduce = None
        loss_nodes.backward(gradient=loss_nodes.new_full(loss_nodes.shape, 1.0 / n) if n > 0 else torch.zeros_like(loss_nodes))
        return loss_nodes.detach().sum() / max(n, 1.0)
    # (new_full is a fill kernel: torch.tensor(x, device=cuda) would be a synchronous host-to-device copy, i.e.
    # a hidden stream sync between the forward and the backward pass)
    total = loss_nodes.sum()
    stats = torch.stack([total.detach(), loss_nodes.new_full((), float(loss_nodes.shape[0]))])
    if distributed:
        reduce(stats)
    # The early (side-stream, in-place) all-reduce of the interaction blocks' range is only correct when the backward's
    # flat gradient buffer BECOMES the parameters' .grad (AccumulateGrad steals the views when .grad is None).  With a
    # .grad already present -- zero_grad(set_to_none=False), micro-batch accumulation, a preset .grad -- AccumulateGrad
    # runs `p.grad += view` on the main stream while the side stream reduces the same memory, and the gather-scatter
    # branch below would reduce p.grad a second time: arm it only when every trainable parameter's .grad is None.
    params = [p for p in model.parameters() if p.requires_grad]
    arm = distributed and overlap and all(p.grad is None for p in params)
    model._dp_early_reduce = reduce if arm else None
    try:
        (total / stats[1]).backward()
    finally:
        model._dp_early_reduce = None
    if distributed:
        flat = getattr(model, "_flat_grad", None)
        if flat is not None and all(p.grad is not None and p.grad.untyped_storage().data_ptr() ==
                                    flat.untyped_storage().data_ptr() for p in params) and \
                sum(p.numel() for p in params) == flat.numel():
            # the fused training step hands out views of ONE flat gradient buffer: reduce it in place
            early = getattr(model, "_dp_early_done", None)
            if early is not None:  # the blocks' range is already on its way on the side stream: head and tail now
                side, off, cnt = early
                if off > 0:
                    reduce(flat[:off])
                if off + cnt < flat.numel():
                    reduce(flat[off + cnt:])
                torch.cuda.current_stream(flat.device).wait_stream(side)
                model._dp_early_done = None
                model._last_reduce = "flat-in-place, blocks early"
            else:
                reduce(flat)
                model._last_reduce = "flat-in-place"
        else:
            early = getattr(model, "_dp_early_done", None)
            if early is not None:
                # (cannot happen when the early reduce is armed as above; if a caller armed it by hand the reduced range
                # must not be reduced again: fail loudly rather than double-count 83 % of the gradient)
                torch.cuda.current_stream(early[0].device).wait_stream(early[0])
                model._dp_early_done = None
                raise RuntimeError("tsdiff_amd.dp_backward: the interaction blocks' gradients were all-reduced early, but "
                                   "the parameters' .grad are not views of the step's flat gradient (a .grad existed "
                                   "before backward?): the gradients of this step are inconsistent -- zero_grad(set_to_none="
                                   "True) before get_loss, or do not set model._dp_early_reduce by hand")
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            reduce(flat)
            off = 0
            for p in params:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
            model._last_reduce = "gather-scatter"
    return stats[0] / stats[1]
