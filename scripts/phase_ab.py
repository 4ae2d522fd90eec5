#!/usr/bin/env python3
"""Diagnostic (GPU box): per-phase shader cycles of the single-end search kernel (URMAPX_PHASE_STATS, diagnostic
instantiation with phase 6 inline), for whichever build URMAPX_LIB names.  usage: phase_ab.py [genome_mbp] [n_reads]"""
import os
import sys

os.environ["URMAPX_PHASE_STATS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
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
wl = bench.Workload(torch, api, dev, d_seq, lens, offs, False, 150, 0.01, 0.001, n, 2, 4242)
dt, kms = wl.timed([m], 2, 1)
pc = m.phase_cycles()
names = ("setup", "phase1+2", "phase3", "walks(+probe)", "phase4+5", "-", "phase6", "output", "locate+fetch", "compare", "xdrop", "ordered")
print(os.environ.get("URMAPX_LIB", "default lib"), f"kernels ms {kms}")
print("  kcycles per read: " + ", ".join(f"{a} {c / n / 1e3:.2f}" for a, c in zip(names, pc)) + f"; sum of the first eight {sum(pc[:8]) / n / 1e3:.1f}")
