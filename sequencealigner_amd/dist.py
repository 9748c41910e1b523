"""Process-per-GPU plumbing for multi-GPU runs.

The hot path has no exchange step: pairs and sequences are independent (the reference farms them out
one task per read, src/mapper.cpp:662-679), so ranks only need (1) a deterministic split of the units,
(2) barriers around the timed region and (3) a MAX/SUM reduction of a few scalars.  torch.distributed
is used for exactly that (backend nccl = RCCL on the GPU box, gloo on CPU).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


class World:
    def __init__(self, rank, size, local_rank, backend):
        self.rank, self.size, self.local_rank, self.backend = rank, size, local_rank, backend

    @property
    def is_dist(self):
        return self.size > 1


def init(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (set by torch.distributed.run).
    SA_DIST_BACKEND=gloo forces the CPU backend on a GPU box: the way to rehearse several ranks on a box with
    one GPU, where RCCL refuses two ranks on the same device (the ranks then share device 0, see device_of)."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if size > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=size)
    return World(rank, size, local_rank, backend)


def device_of(world):
    """Index of the GPU this rank computes on: its local rank; under gloo on a box with fewer GPUs than ranks
    (a rehearsal) the ranks wrap around the GPUs there are."""
    if not world.is_dist:
        return 0
    if world.backend == "gloo":
        return world.local_rank % max(torch.cuda.device_count(), 1)
    return world.local_rank


def barrier(world):
    if world.is_dist:
        dist.barrier()


def _reduce(world, value, op):
    if not world.is_dist:
        return float(value)
    dev = torch.device("cuda", world.local_rank) if world.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=op)
    return float(t.item())


def reduce_max(world, value):
    return _reduce(world, value, dist.ReduceOp.MAX)


def reduce_sum(world, value):
    return _reduce(world, value, dist.ReduceOp.SUM)


def finalize(world):
    if world.is_dist and dist.is_initialized():
        dist.destroy_process_group()


def shard_by_cost(costs, n_shards):
    """Greedy longest-processing-time split of units with the given costs (e.g. query_len*target_len
    per pair, bases per sequence) into n_shards index arrays of near-equal total cost.  Deterministic;
    every rank computes the same answer and keeps its own shard."""
    costs = np.asarray(costs, dtype=np.int64)
    n = costs.shape[0]
    if n_shards <= 1:
        return [np.arange(n, dtype=np.int64)]
    if n and costs.min() == costs.max():  # uniform: contiguous equal blocks
        bounds = np.linspace(0, n, n_shards + 1).astype(np.int64)
        return [np.arange(bounds[i], bounds[i + 1], dtype=np.int64) for i in range(n_shards)]
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(n_shards, dtype=np.int64)
    owner = np.empty(n, dtype=np.int64)
    # LPT in blocks: exact for the largest units, round-robin over sorted blocks for the tail
    head = min(n, 4096)
    for idx in order[:head]:
        s = int(np.argmin(load))
        owner[idx] = s
        load[s] += costs[idx]
    tail = order[head:]
    if tail.size:
        ranks = np.argsort(load, kind="stable")
        owner[tail] = ranks[np.arange(tail.size) % n_shards]
    return [np.sort(np.nonzero(owner == s)[0]) for s in range(n_shards)]
