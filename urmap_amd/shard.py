"""Read sharding across the GPUs of a node: batch b of the input goes to rank b mod world.
Mapping is per-read independent (map.cpp:11-25 hands each read to whichever thread asks next), so there is no
exchange step and no collective: each rank maps its batches against its own replica of the index."""
from __future__ import annotations


def batches_for_rank(n_reads: int, batch: int, rank: int, world: int):
    """[(lo, hi)) read ranges this rank maps, in input order."""
    out = []
    b = 0
    lo = 0
    while lo < n_reads:
        hi = min(n_reads, lo + batch)
        if b % world == rank:
            out.append((lo, hi))
        lo = hi
        b += 1
    return out
