"""CPU suite: the multi-process plumbing used by bench.py --gpus N, exercised with 2 gloo ranks.
The data path has no collective (pairs are independent); ranks only agree on a split, meet at barriers
and reduce a few scalars."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, q):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world_size), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from sequencealigner_amd import dist, synth
    world = dist.init(backend="gloo")
    assert world.size == world_size and world.rank == rank and world.is_dist
    # every rank derives the same split and keeps its own shard
    qlen = np.random.default_rng(5).integers(50, 300, size=999)
    tlen = np.random.default_rng(6).integers(100, 600, size=999)
    shards = dist.shard_by_cost(qlen * tlen, world.size)
    mine = shards[world.rank]
    cells = float((qlen[mine] * tlen[mine]).sum())
    # stand-in for the timed region: rank-dependent "elapsed" values
    dist.barrier(world)
    elapsed = 1.0 + rank
    total_cells = dist.reduce_sum(world, cells)
    slowest = dist.reduce_max(world, elapsed)
    dist.barrier(world)
    # weak-scaling inputs differ per rank (seeded by rank) but have the same shape
    q, t = synth.pairs_short_read_window(16, seed=1000 + rank)
    out = (rank, int(len(mine)), total_cells, slowest, int(np.asarray(q).sum()))
    dist.finalize(world)
    return out


def _entry(rank, world_size, port, queue):
    queue.put(_worker(rank, world_size, port, None))


def test_two_ranks_agree_on_split_and_reductions():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_entry, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(queue.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    qlen = np.random.default_rng(5).integers(50, 300, size=999)
    tlen = np.random.default_rng(6).integers(100, 600, size=999)
    assert res[0][1] + res[1][1] == 999
    assert res[0][2] == res[1][2] == float((qlen * tlen).sum())   # SUM over ranks
    assert res[0][3] == res[1][3] == 2.0                          # MAX over ranks
    assert res[0][4] != res[1][4]                                 # per-rank inputs differ


def test_bench_split_of_the_semiglobal_set_under_two_gloo_ranks():
    """bench.py --gpus 2 as the driver starts it (torch.distributed.run, one process per rank), with the CPU
    backend and --split-only: no GPU work, only the logic that deals the fixed semi-global set to the ranks.  The
    line must call the leg's scaling "strong", and the ranks' shares must add up to --sg-total and tile it."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SA_DIST_BACKEND="gloo")
    for total in (1001, 7):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2",
               "--split-only", "--sg-total", str(total)]
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [x for x in res.stdout.splitlines() if x.startswith("{")]
        assert len(lines) == 1          # rank 0 alone prints
        leg = json.loads(lines[0])["legs"]["sg_10kx11k"]
        assert json.loads(lines[0])["n_gpus"] == 2
        assert leg["scaling"] == "strong"
        assert leg["pairs_all_ranks"] == total == leg["sg_total"]
        assert leg["first_pair_this_rank"] == 0 and leg["end_of_last_shard"] == total
        assert abs(leg["pairs_this_rank"] - total / 2) <= 1
