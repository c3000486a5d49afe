#!/usr/bin/env python3
"""BASELINE configs[2] harness: a reaction test set x an M-checkpoint ensemble, reactions sharded over the
GPUs of one node (one process per GPU, no data-path collective), the way `sampling.py` batches them
(`batch_size` reactions per call of `dynamic_sampling`, sampling.py:169-231).

    python tools/sample_sharded.py --graphs 2400 --models 8 --batch-size 100 --steps 5000
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/sample_sharded.py ...

Synthetic reactions and closed-form weights (the wb97xd3 pickles and trained checkpoints are LFS blobs absent
from the reference tree); real use replaces `make_graphs` by the unpickled test set and `make_models` by
`tsdiff_amd.io.load_checkpoint` + `get_model(ckpt["config"].model)` + `load_state_dict(ckpt["model"])`.
Rank 0 prints one JSON line and (with --out DIR) writes DIR/samples_all.pkl in the reference's result format
(`sampling.py:218-243`: a list of PyG `Data` records, one per reaction, with `pos_gen`; tsdiff_amd.io).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tsdiff_amd import synth  # noqa: E402
from tsdiff_amd.distributed import sample_sharded  # noqa: E402
from tsdiff_amd.epsnet import get_model  # noqa: E402
from tsdiff_amd.sampler import EnsembleSampler  # noqa: E402
from tsdiff_amd.utils import AttrDict  # noqa: E402


def make_graphs(n, seed):
    b = synth.wb97xd3_like_batch(n, seed=seed)
    off = np.concatenate([[0], np.cumsum(b["num_nodes_per_graph"])])
    gs = []
    for g in range(n):
        lo, hi = off[g], off[g + 1]
        sel = (b["bond_index"][0] >= lo) & (b["bond_index"][0] < hi)
        gs.append({k: torch.from_numpy(b[k][lo:hi]) for k in ("atom_type", "r_feat", "p_feat", "pos")}
                  | {"bond_index": torch.from_numpy(b["bond_index"][:, sel] - lo),
                     "bond_type": torch.from_numpy(b["bond_type"][sel])})
    return gs


def collate(gs, dev):
    npg = [int(g["atom_type"].shape[0]) for g in gs]
    off = np.concatenate([[0], np.cumsum(npg)])
    out = {k: torch.cat([g[k] for g in gs]).to(dev) for k in ("atom_type", "r_feat", "p_feat", "pos", "bond_type")}
    out["bond_index"] = torch.cat([g["bond_index"] + int(off[i]) for i, g in enumerate(gs)], dim=1).to(dev)
    out["batch"] = torch.repeat_interleave(torch.arange(len(gs)), torch.tensor(npg)).to(dev)
    return out, npg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=400)
    ap.add_argument("--models", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=100)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--sampling-type", default="ld")
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=2022,
                    help="run seed (sampling.py:82 default): rank r draws its initial positions and its Langevin noise "
                         "from the streams seeded `seed ^ r` (SURVEY 8e), so a run is reproducible for a (seed, GPU "
                         "count) pair and no two ranks share a stream")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    cfg = synth.DEFAULT_MODEL_CONFIG
    models = []
    for m in range(args.models):
        model = get_model(AttrDict(cfg))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, m).items()}, strict=False)
        models.append(model.to(dev))
    sampler = EnsembleSampler(models)
    graphs = make_graphs(args.graphs, seed=7)  # the same list on every rank

    def sample_fn(shard, r):
        res = []
        gen = torch.Generator(device=dev)
        gen.manual_seed(args.seed ^ r)  # initial positions: this rank's own stream
        for k, s in enumerate(range(0, len(shard), args.batch_size)):  # sampling.py:169 batching
            b, npg = collate(shard[s:s + args.batch_size], dev)
            pos_init = torch.randn(b["pos"].shape[0], 3, device=dev, generator=gen)  # sampling.py:190
            pos, _ = sampler.dynamic_sampling(b["atom_type"], b["r_feat"], b["p_feat"], pos_init, b["bond_index"],
                                              b["bond_type"], b["batch"], len(npg), extend_order=True,
                                              n_steps=args.steps, step_lr=1e-7, clip=1000,
                                              sampling_type=args.sampling_type, return_traj=False,
                                              # device Philox key of this (rank, batch): no overlap across either
                                              seed=((args.seed ^ r) << 20) + k)
            res += list(torch.split(pos.cpu(), npg))  # sampling.py:218-223
        return res

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sample_sharded(graphs, sample_fn)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        atoms = sum(int(g["atom_type"].shape[0]) for g in graphs)
        assert len(res) == len(graphs) and all(r.shape == g["pos"].shape for r, g in zip(res, graphs))
        print(json.dumps({"graphs": args.graphs, "atoms": atoms, "checkpoints": args.models, "gpus": world,
                          "seed": args.seed,
                          "steps": args.steps, "seconds": round(dt, 3),
                          "atoms_steps_per_s": round(atoms * args.steps / dt, 1),
                          "checkpoint_forwards_per_s": round(
                              -(-args.graphs // (args.batch_size * world)) * world * args.models * args.steps / dt, 1)}))
        if args.out:
            from tsdiff_amd import io as tio
            w = tio.ResultWriter(args.out)
            recs = [tio.SampleRecord(atom_type=g["atom_type"], r_feat=g["r_feat"], p_feat=g["p_feat"], pos=g["pos"],
                                     edge_index=g["bond_index"], edge_type=g["bond_type"], smiles=f"synthetic_{k}",
                                     pos_gen=r) for k, (g, r) in enumerate(zip(graphs, res))]
            w.add_batch(recs)
            print("wrote", w.finish())
    if "RANK" in os.environ:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
