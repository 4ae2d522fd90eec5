#!/usr/bin/env python3
"""Diagnostic (GPU box): where the single-end search kernel's cycles go, per phase and per read.
usage: URMAPX_PHASE_STATS=1 python3 scripts/read_cost.py [genome_mbp] [read_len] [sub] [indel] [n_reads]
Prints the phase cycle shares, the per-read cycle distribution and what the oracle says about the costliest reads."""
import os
import sys

os.environ.setdefault("URMAPX_PHASE_STATS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import bench
import oracle_lib as ol
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 200
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sub = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
indel = float(sys.argv[4]) if len(sys.argv) > 4 else 0.001
n = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
m = api.Mapper(index, device=0)
wl = bench.Workload(torch, api, dev, d_seq, lens, offs, False, L, sub, indel, n, 2, 4242)
dt, kms = wl.timed([m], 2, 1)
print(f"genome {mbp} Mbp {desc}\nL={L} sub={sub} indel={indel} n={n}: kernels ms {kms}, {2 * n / dt:.0f} reads/s")
pc = m.phase_cycles()
sub8, pc = pc[8:], pc[:8]
tot = max(1, sum(pc))
names = ("setup", "phase1+2", "phase3", "chain walks", "phase4", "phase5", "phase6", "output")
print("phase cycle shares: " + ", ".join(f"{a} {100.0 * c / tot:.1f}%" for a, c in zip(names, pc)) + f"; cycles/read {tot / n:.0f}")
print("inside candidate batches: " + ", ".join(f"{a} {100.0 * c / tot:.1f}%" for a, c in zip(("locate+fetch", "compare", "xdrop", "ordered"), sub8)))
cyc = m.read_cycles(n)
srt = np.sort(cyc)
print("cycles per read: mean %.0f, median %d, p90 %d, p99 %d, p99.9 %d, max %d; top 1%% of reads hold %.1f%% of the cycles, top 0.1%% %.1f%%" % (
    cyc.mean(), srt[n // 2], srt[int(n * 0.9)], srt[int(n * 0.99)], srt[int(n * 0.999)], srt[-1],
    100.0 * srt[-n // 100:].sum() / cyc.sum(), 100.0 * srt[-max(1, n // 1000):].sum() / cyc.sum()))
oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
par, cnt, _ = wl.check(oi, n, bench.host_cores())
print("parity", par, "\nwork/read", {k: round(v, 1) for k, v in cnt.items()})
hb = wl.last[: n * L].cpu().numpy()
ho = np.arange(n + 1, dtype=np.uint64) * L
ores, _, _ = oi.map_se(hb, ho, threads=bench.host_cores())
order = np.argsort(-cyc)
print("costliest reads: cycles, oracle hsp_count, hit_count, exit_phase, score")
for i in order[:15]:
    print("  ", int(cyc[i]), int(ores["hsp_count"][i]), int(ores["hit_count"][i]), int(ores["exit_phase"][i]), int(ores["score"][i]))
for name in ("hsp_count", "hit_count"):
    x = ores[name].astype(np.float64)
    print(f"corr(cycles, {name}) = {np.corrcoef(cyc, x)[0, 1]:.3f}; mean {x.mean():.2f}")
bins = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 100000]
h = ores["hsp_count"]
for lo, hi in zip(bins[:-1], bins[1:]):
    sel = (h >= lo) & (h < hi)
    if sel.any():
        print(f"  hsp_count [{lo},{hi}): {sel.mean() * 100:.2f}% of reads, {100.0 * cyc[sel].sum() / cyc.sum():.1f}% of cycles, mean cycles {cyc[sel].mean():.0f}")
