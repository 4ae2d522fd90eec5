#!/usr/bin/env python3
"""Diagnostic (GPU box): single-end search kernel time with the schedule cut after step N (URMAPX_DEBUG_STOP), one
process, one index.  usage: stop_sweep.py [genome_mbp] [read_len] [sub] [indel] [n_reads] [stops...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 800
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
indel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
n = int(sys.argv[5]) if len(sys.argv) > 5 else 1000000
stops = [int(x) for x in sys.argv[6:]] or [97, 98, 99, 100, 1, 3, 104, 401, 402, 403, 4, 0]
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
m = api.Mapper(index, device=0)
wl = bench.Workload(torch, api, dev, d_seq, lens, offs, False, L, sub, indel, n, 3, 4242)
print(f"genome {mbp} Mbp, L={L} sub={sub} indel={indel} n={n}")
os.environ.pop("URMAPX_PHASE_STATS", None)
os.environ.pop("URMAPX_DEBUG_STOP", None)
dt, kms = wl.timed([m], 3, 1)
print(f"production kernel: probe {kms[0]:.2f} ms, search {kms[1]:.2f} ms; stages (main, dp, finalize, main2, dp2, finalize2, general) " + ", ".join(f"{x:.2f}" for x in m.stage_ms()) + f"; dp stats {m.dp_stats()}")
sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("SWEEP_CHECK"):
    import oracle_lib as ol
    oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
    par, cnt, _ = wl.check(oi, int(os.environ["SWEEP_CHECK"]), bench.host_cores())
    print("parity", par)
for s in stops:
    os.environ["URMAPX_DEBUG_STOP"] = str(s)
    dt, kms = wl.timed([m], 3, 1)
    print(f"stop {s:4d}: search {kms[1]:8.2f} ms")
