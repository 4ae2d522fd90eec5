#!/usr/bin/env python3
"""Diagnostic (GPU box): how much of the search kernel's duration is tail (persistent waves, heavy-tailed cost per read)?
Per-read cycles from the diagnostic kernel, then the production kernel on the same batch in input order, with the
costliest reads first (longest-processing-time order) and with them last (worst case)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
indel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
n = 1000000
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
m = api.Mapper(index, device=0)
wl = bench.Workload(torch, api, dev, d_seq, lens, offs, False, L, sub, indel, n, 1, 4242)
dt, kms = wl.timed([m], 3, 1)
print(f"input order: search stages {[round(x, 2) for x in wl.stage_ms]}")
os.environ["URMAPX_PHASE_STATS"] = "1"
wl.timed([m], 1, 0)
cyc = m.read_cycles(n)
os.environ.pop("URMAPX_PHASE_STATS")
srt = np.sort(cyc)
print("diagnostic kernel cycles per read: mean %.0f median %d p99 %d p99.9 %d max %d" % (cyc.mean(), srt[n // 2], srt[int(n * .99)], srt[int(n * .999)], srt[-1]))
batch = wl.batches[0].reshape(n, L)
for name, order in (("costliest first", np.argsort(-cyc, kind="stable")), ("costliest last", np.argsort(cyc, kind="stable"))):
    wl.batches[0] = batch[torch.from_numpy(order.copy()).to(dev)].reshape(-1).contiguous()
    dt, kms = wl.timed([m], 3, 1)
    print(f"{name}: search stages {[round(x, 2) for x in wl.stage_ms]}")
