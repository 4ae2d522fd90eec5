#!/usr/bin/env python3
"""Diagnostic (GPU box): (1) the single-end step against the batch size -- T(n) = a + b n says what a launch sequence costs whatever it maps (ramp, tail, the second pass's empty
launches); (2) whole batches alternating over TWO mapping contexts of the device (what two lanes of urmapx_map_files do: context B's search kernel fills the CUs that context A's
tail and phase-6 launches leave idle) against one context running its steps back to back.
usage: r6_pingpong.py [genome_mbp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000):
    m = api.Mapper(index, device=0)
    wl = bench.Workload(torch, api, dev, d_seq, lens, offs, False, 150, 0.01, 0.001, n, 3, 4242)
    dt, kms = wl.timed([m], 6, 2)
    print(f"one context, {n} reads per step: step {1e3 * dt / 6:.3f} ms, search {wl.stage_ms[0]:.3f}, dp {wl.stage_ms[1]:.3f}, finalize {wl.stage_ms[2]:.3f} -> {6 * n / dt / 1e6:.2f} M reads/s", flush=True)
    m.close()
    del wl
    torch.cuda.empty_cache()
n = 1_000_000
for pe, L, name in ((False, 150, "se150"), (True, 150, "pe2x150"), (False, 250, "se250")):
    sub, indel = (0.01, 0.001) if L == 150 else (0.04, 0.01)
    ms = [api.Mapper(index, device=0) for _ in range(2)]
    wls = [bench.Workload(torch, api, dev, d_seq, lens, offs, pe, L, sub, indel, n, 3, 4242 + 100 * k) for k in range(2)]
    torch.cuda.synchronize()
    for mode in ("one context", "two contexts, alternating", "one context", "two contexts, alternating"):
        K, W = 12, 2
        for k in range(W):
            wls[k % 2].step([ms[k % 2]], k)
        for m in ms:
            m.sync()
        t0 = time.perf_counter()
        for k in range(K):
            c = k % 2 if mode.startswith("two") else 0
            if mode.startswith("two"):
                pass  # a context's launches are ordered on its own stream: batch k + 2 queues behind batch k
            wls[c].step([ms[c]], k)
            if not mode.startswith("two"):
                ms[c].sync()
        for m in ms:
            m.sync()
        dt = time.perf_counter() - t0
        print(f"{name}, {mode}: {1e3 * dt / K:.3f} ms per 1 M-read batch -> {K * n / dt / 1e6:.2f} M reads/s", flush=True)
    # the results of the alternating run are the oracle's?  (the last batch of each context against its own one-context run)
    for m in ms:
        m.close()
    del wls
    torch.cuda.empty_cache()
