#!/usr/bin/env python3
"""Diagnostic (GPU box): paired-end search kernel time with the schedule cut after stage N (URMAPX_DEBUG_STOP_PE): 1 setup +
enumeration + seed gather, 2 + pairing loop, 3 + all-seeds pass, 4 + pending stage, 5 everything but the rescue DP, 0 all.
usage: pe_sweep.py [genome_mbp] [n_reads]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
m = api.Mapper(index, device=0)
wl = bench.Workload(torch, api, dev, d_seq, lens, offs, True, 150, 0.01, 0.001, n, 3, 4242)
for s in (0, 1, 2, 3, 41, 42, 43, 4, 5, 0):
    if s:
        os.environ["URMAPX_DEBUG_STOP_PE"] = str(s)
    else:
        os.environ.pop("URMAPX_DEBUG_STOP_PE", None)
    dt, kms = wl.timed([m], 3, 1)
    print(f"stop {s}: probe {kms[0]:.2f} ms, search {kms[1]:.2f} ms", flush=True)
