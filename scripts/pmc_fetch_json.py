#!/usr/bin/env python3
"""pmc_summary.py output (FETCH_SIZE pass) -> the profiles/rN/pmc_fetch_hg38scale_*.json format bench.py reads.
usage: pmc_fetch_json.py raw.json out.json read_len [note]"""
import json, sys
raw = json.load(open(sys.argv[1]))
L = int(sys.argv[3])
out = {"source": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                 + (sys.argv[4] if len(sys.argv) > 4 else "") + " (hg38-scale workload, 1M reads per launch), scripts/profile_fullscale.sh",
       "unit_note": "FETCH_SIZE is reported in KiB; for this random 64-byte-sector access pattern it matches the known byte count of the "
                    "probe kernel (254 k-mers x 64 B + 6 % line-straddling slots + the read = 16.5 KB per 150 bp read), so no gfx950 "
                    "half-count correction applies (that correction is for wide coalesced 128-B requests, MI355X_MICROARCH.md HBM section)",
       "reads_per_launch": 1000000, "read_len": L, "kernels": {}}
for k, v in raw.items():
    kib = v["FETCH_SIZE"]["avg"]
    out["kernels"][k] = {"launches": v["FETCH_SIZE"]["dispatches"], "FETCH_SIZE_KiB_avg": kib, "hbm_read_bytes_per_launch": kib * 1024.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
