#!/usr/bin/env python3
"""Diagnostic (GPU box): kernel time against resident waves -- the single-end and the pair search kernels with their grids cut to
B blocks per CU (URMAPX_TEST_BLOCKS_PER_CU; one wave per block, 4 SIMDs per CU: B / 4 waves per SIMD).  The slope from 8 to 12 to
16 blocks says what a wave more per SIMD is worth before its registers are paid for.
usage: r6_waves_sweep.py [genome_mbp] [n_reads]"""
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
for pe, L, name in ((False, 150, "se150"), (True, 150, "pe2x150"), (False, 250, "se250")):
    wl = bench.Workload(torch, api, dev, d_seq, lens, offs, pe, L, 0.01 if L == 150 else 0.04, 0.001 if L == 150 else 0.01, n, 3, 4242)
    for b in (4, 8, 12, 16, 0):
        if b:
            os.environ["URMAPX_TEST_BLOCKS_PER_CU"] = str(b)
        else:
            os.environ.pop("URMAPX_TEST_BLOCKS_PER_CU", None)
        m = api.Mapper(index, device=0)
        dt, kms = wl.timed([m], 3, 1)
        st = "" if pe else f" stages {[round(float(x), 2) for x in wl.stage_ms[:3]]}"
        print(f"{name} blocks/CU {b or 'all'}: search {kms[1]:.2f} ms, step {1e3 * dt / 3:.2f} ms{st}", flush=True)
        m.close()
    del wl
    torch.cuda.empty_cache()
