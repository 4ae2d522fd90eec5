#!/usr/bin/env python3
"""pmc_summary.py output (FETCH_SIZE or WRITE_SIZE pass) -> the profiles/rN/pmc_fetch_*.json / pmc_write_*.json format bench.py reads.
usage: pmc_fetch_json.py raw.json out.json mode(se|pe) read_len genome_bp code_version steps [note] [FETCH_SIZE|WRITE_SIZE]
`steps` = bench steps the profiled run executed (warm-up included): kernels launched several times per step (dp_kernel:
one launch per round and pass) are reported per STEP, the others per launch."""
import json, sys
raw = json.load(open(sys.argv[1]))
mode, L, genome_bp, code, steps = sys.argv[3], int(sys.argv[4]), float(sys.argv[5]), sys.argv[6], int(sys.argv[7])
counter = sys.argv[9] if len(sys.argv) > 9 else "FETCH_SIZE"
field = "hbm_read_bytes_per_launch" if counter == "FETCH_SIZE" else "hbm_write_bytes_per_launch"
out = {"source": "rocprofv3 --pmc " + counter + " -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e "
                 + (sys.argv[8] if len(sys.argv) > 8 else "") + " (hg38-scale workload, 1M reads per launch), scripts/r6_profile2.sh (r6_profile.sh in the first half of round 6, r5_profile.sh in round 5)",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are reported in KiB; for this random 64-byte-sector access pattern it matches the known byte count of the "
                    "probe kernel (k-mers x 64 B + 6 % line-straddling slots + the read), so no gfx950 half-count correction applies "
                    "(that correction is for wide coalesced 128-B requests, MI355X_MICROARCH.md HBM section)",
       "mode": mode, "read_len": L, "genome_bp": genome_bp, "code_version": code, "reads_per_launch": 1000000, "kernels": {}}
for k, v in raw.items():
    if counter not in v:
        continue
    f = v[counter]
    per = f["sum"] / steps if f["dispatches"] > steps else f["avg"]
    out["kernels"][k] = {"launches": f["dispatches"], "launches_per_step": round(f["dispatches"] / steps, 2), counter + "_KiB": per,
                         field: per * 1024.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
