"""CPU, world_size 2 (gloo): the multi-GPU path is "reads sharded by rank, index replicated, no collective on
the data path" (SURVEY.md 8e).  This checks the sharding helper partitions a read set exactly and that the
rank-0 aggregate used for reporting sees every read once."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, n_reads, batch):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from urmap_amd import shard
    mine = shard.batches_for_rank(n_reads, batch, rank, world)
    seen = torch.zeros(n_reads, dtype=torch.int32)
    total = 0
    for lo, hi in mine:
        seen[lo:hi] += 1
        total += hi - lo
    dist.all_reduce(seen)
    t = torch.tensor([total], dtype=torch.int64)
    dist.all_reduce(t)
    assert int(t.item()) == n_reads
    assert bool((seen == 1).all())
    # max-over-ranks timing reduction used by bench.py
    x = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    assert x.item() == float(world)
    dist.destroy_process_group()


def test_read_sharding_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, 100003, 4096), nprocs=2, join=True)


def test_batches_cover_everything_single_rank():
    from urmap_amd import shard
    b = shard.batches_for_rank(10, 4, 0, 1)
    assert b == [(0, 4), (4, 8), (8, 10)]
    assert shard.batches_for_rank(10, 4, 1, 3) == [(4, 8)]
    assert shard.batches_for_rank(0, 4, 0, 2) == []
