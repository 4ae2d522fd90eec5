#!/usr/bin/env python3
"""GPU box: parity fuzz on one resident index -- several read lengths, error rates and seeds, single-end and paired-end,
every field + path against the oracle.  usage: fuzz_gpu.py [genome_mbp] [reads_per_case]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import bench
import oracle_lib as ol
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 800
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 777, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
ms = [api.Mapper(index, device=0), api.Mapper(index, device=0)]
oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
cores = bench.host_cores()
bad = 0
cases = [(False, 150, 0.01, 0.001), (False, 150, 0.03, 0.01), (False, 100, 0.02, 0.004), (False, 250, 0.04, 0.01), (False, 250, 0.01, 0.002),
         (False, 300, 0.03, 0.008), (False, 64, 0.01, 0.0), (True, 150, 0.01, 0.001), (True, 100, 0.03, 0.005), (True, 250, 0.02, 0.004),
         (False, 500, 0.02, 0.004), (False, 1000, 0.01, 0.002), (True, 270, 0.03, 0.006),  # round 3: the 512 / 1024-base kernel classes, the longest pairs
         (False, 151, 0.02, 0.004), (False, 152, 0.02, 0.004), (False, 128, 0.03, 0.006), (True, 151, 0.02, 0.004)]  # round 6: either side of the two-chunk instance's 128 k-mer starts, the <2> class with its row store in LDS
n_all = n
for ci, (pe, L, sub, indel) in enumerate(cases):
    n = n_all if L <= 300 else n_all // 4
    for streams in ((1, 2) if ci < 2 else (1,)):
        wl = bench.Workload(torch, api, dev, d_seq, lens, offs, pe, L, sub, indel, n, 1, 9000 + 31 * ci, streams=streams)
        dt, kms = wl.timed(ms[:streams], 1, 0)
        par, cnt, t = wl.check(oi, n, cores)
        ok = par["bit_identical_to_oracle"]
        bad += 0 if ok else 1
        print(f"{'PE' if pe else 'SE'} L={L} sub={sub} indel={indel} streams={streams}: {n / dt / 1e6:.1f} M reads/s, "
              f"{'OK' if ok else 'MISMATCH ' + str(par.get('mismatches'))}, gapped paths {par['gapped_paths_checked']}, mapped {par['mapped_frac']}", flush=True)
        del wl
        torch.cuda.empty_cache()
print("FUZZ", "FAILED" if bad else "PASSED", bad)
