"""One process per GPU: read batches sharded by rank, index replicated, no collective on the data path.

The reference fans reads over the threads of one process (map.cpp:58-61: `#pragma omp parallel num_threads(ThreadCount)`,
each thread pulling the next read from the shared source, seqsource.cpp:30-66).  Here the unit that pulls work is a
rank = one GPU; what ranks share is nothing but the input order, so the only communication is the launch itself, a
barrier and a max-over-ranks of the wall time for reporting (torch.distributed: RCCL when every rank has its own GPU,
gloo when ranks are made to share a device for testing or have no GPU at all).

Used by bench.py (launch, init, barrier, reductions) and by the world-size-2 tests.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def batches_for_rank(n_reads: int, batch: int, rank: int, world: int):
    """[(lo, hi)) read ranges this rank maps, in input order: batch b of the input goes to rank b mod world."""
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


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched() -> bool:
    """True inside a rank started by torch.distributed.run (the driver's launch or launch_ranks below)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def launch_ranks(script: str, argv: list[str], n: int, n_devices: int | None = None, timeout: float | None = None) -> int:
    """Start `n` ranks of `script argv...` as CHILD processes through torch.distributed.run on this node and return the
    launcher's exit code.  The calling process must not have touched the GPU (it only counts devices; it never execs).
    With fewer visible devices than ranks the children are told to share them (URMAP_RANK_DEVICES) and use gloo."""
    if n_devices is None:
        try:
            import torch
            n_devices = torch.cuda.device_count()  # counts without initialising the GPU runtime
        except Exception:
            n_devices = 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    if n_devices < n:
        env["URMAP_RANK_DEVICES"] = str(max(n_devices, 0))
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


class Ranks:
    """rank / world / device of this process and the three collectives the bench contract needs."""

    def __init__(self, want_gpu=True):
        self.rank = int(os.environ.get("RANK", 0))
        self.world = int(os.environ.get("WORLD_SIZE", 1))
        self.local_rank = int(os.environ.get("LOCAL_RANK", 0))
        shared = os.environ.get("URMAP_RANK_DEVICES")  # set by launch_ranks when ranks outnumber devices
        forced = os.environ.get("URMAP_BENCH_FORCE_DEVICE")  # older spelling: every rank on this one device
        if forced is not None:
            self.device_index, self.shared = int(forced), self.world > 1
        elif shared is not None:
            nd = int(shared)
            self.device_index, self.shared = (self.local_rank % nd if nd > 0 else -1), True
        else:
            self.device_index, self.shared = self.local_rank, False
        self.backend = None
        self.dist = None
        self.want_gpu = want_gpu

    def init(self, torch):
        if self.want_gpu:
            torch.cuda.set_device(self.device_index)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # the other ranks sit in the first collective while rank 0 builds the index (24 s at hg38 scale, more on a slow host):
            # the wait is bounded explicitly, not by a backend's default
            import datetime
            limit = datetime.timedelta(seconds=int(os.environ.get("URMAP_RANK_TIMEOUT_S", 1800)))
            if self.shared or not self.want_gpu:
                self.backend = "gloo"
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world, timeout=limit)
            else:
                self.backend = "nccl"  # = RCCL on ROCm
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, timeout=limit,
                                        device_id=torch.device("cuda", self.device_index))
            self.dist = dist
        return self

    def barrier(self, torch=None):
        if self.dist is not None:
            self.dist.barrier()
        if torch is not None and self.want_gpu:
            torch.cuda.synchronize()

    def max_over_ranks(self, torch, x: float) -> float:
        if self.dist is None:
            return x
        dev = "cpu" if self.backend == "gloo" else torch.device("cuda", self.device_index)
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, torch, x: float) -> float:
        if self.dist is None:
            return x
        dev = "cpu" if self.backend == "gloo" else torch.device("cuda", self.device_index)
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_floats(self, torch, xs):
        """every rank's list of floats -> [[rank 0's], [rank 1's], ...] on every rank (reporting only: per-rank step and set-up times)"""
        if self.dist is None:
            return [[float(x) for x in xs]]
        dev = "cpu" if self.backend == "gloo" else torch.device("cuda", self.device_index)
        mine = torch.tensor([float(x) for x in xs], device=dev, dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[float(v) for v in t.tolist()] for t in out]

    def broadcast_bytes(self, torch, t, src=0, chunk=1 << 30):
        """Broadcast a 1-D uint8 tensor in place, in pieces of `chunk` bytes (a 27 GB slot table is one tensor)."""
        if self.dist is None:
            return t
        for lo in range(0, t.numel(), chunk):
            piece = t[lo:lo + chunk]
            if self.backend == "gloo" and piece.is_cuda:  # gloo moves bytes through the host
                h = piece.cpu()
                self.dist.broadcast(h, src=src)
                if self.rank != src:
                    piece.copy_(h)
            else:
                self.dist.broadcast(piece, src=src)
        return t

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
