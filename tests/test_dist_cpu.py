"""CPU, gloo, world size 2: the N>1 sharding used by bench.py (round-robin samples, max-over-ranks
time, summed counters)."""
import os
import socket

import torch.distributed as dist
import torch.multiprocessing as mp

from agc_amd import shard


def _worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.samples_of_rank(n_samples, rank, world)
    seeds = [shard.sample_seed(1000, s, rank, world) for s in range(3)]
    t = shard.reduce_job_time(dist, 1.0 + rank)
    bases, segs = shard.reduce_counters(dist, [100.0 * len(mine), float(len(mine))])
    q.put((rank, mine, seeds, t, bases, segs))
    dist.destroy_process_group()


def test_round_robin_sharding_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, n_samples = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_samples, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    all_samples = sorted(res[0][1] + res[1][1])
    assert all_samples == list(range(n_samples))            # disjoint and complete
    assert set(res[0][2]).isdisjoint(res[1][2])              # no two ranks generate the same sample
    assert res[0][3] == res[1][3] == 2.0                     # job time = slowest rank
    assert res[0][4] == res[1][4] == 700.0 and res[0][5] == 7.0
