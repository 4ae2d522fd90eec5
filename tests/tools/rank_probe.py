"""Started by tests/test_sharding_gloo.py through urmap_amd.ranks.launch_ranks: the rank plumbing bench.py uses
(init, barrier, max- and sum-over-ranks, byte broadcast, batch sharding) on CPU with gloo; rank 0 prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from urmap_amd import ranks  # noqa: E402


def main():
    n_reads, batch = int(sys.argv[1]), int(sys.argv[2])
    R = ranks.Ranks(want_gpu=False).init(torch)
    mine = ranks.batches_for_rank(n_reads, batch, R.rank, R.world)
    seen = torch.zeros(n_reads, dtype=torch.int32)
    for lo, hi in mine:
        seen[lo:hi] += 1
    if R.dist is not None:
        R.dist.all_reduce(seen)
    total = R.sum_over_ranks(torch, float(sum(hi - lo for lo, hi in mine)))
    slowest = R.max_over_ranks(torch, float(R.rank + 1))
    t = torch.arange(5000, dtype=torch.int64).to(torch.uint8) if R.rank == 0 else torch.zeros(5000, dtype=torch.uint8)
    R.broadcast_bytes(torch, t, chunk=1024)
    R.barrier()
    ok_bcast = bool((t == torch.arange(5000, dtype=torch.int64).to(torch.uint8)).all())
    gathered = R.all_gather_floats(torch, [float(R.rank), 10.0 + R.rank])  # per-rank numbers of the N > 1 bench line
    flags = torch.tensor([1.0 if ok_bcast else 0.0])
    all_ok = R.sum_over_ranks(torch, float(flags.item())) == R.world
    if R.rank == 0:
        print(json.dumps({"world": R.world, "backend": R.backend, "total": total, "slowest": slowest,
                          "covered_once": bool((seen == 1).all()), "broadcast_ok": all_ok, "gathered": gathered,
                          "device_index": R.device_index, "shared": R.shared}), flush=True)
    R.close()


if __name__ == "__main__":
    main()
