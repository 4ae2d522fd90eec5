#!/usr/bin/env python3
"""Diagnostic (GPU box): paired-end batch at hg38 scale -- how many reads end with more than 64 hits (the first pass's
list) and what the pair kernel's launches take, for batches of different sizes.  usage: pe_ovf_probe.py [genome_mbp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
m = api.Mapper(index, device=0)
for nb in (1000000, 262144):
    wl = bench.Workload(torch, api, dev, d_seq, lens, offs, True, 150, 0.01, 0.02, nb, 3, 4242)
    dt, kms = wl.timed([m], 3, 1)
    res, ops = wl.results()
    hc = res["hit_count"]
    print(f"nb={nb}: step {1e3 * dt / 3:.2f} ms, kernels {kms}; reads with hit_count > 64: {(hc > 64).sum()}, > 128: {(hc > 128).sum()}, "
          f"== cap-ish (>=256): {(hc >= 256).sum()}, status != 0: {(res['status'] != 0).sum()}, max {hc.max()}")
