#!/bin/bash
# SQ counters of the 250-base workload (search and DP kernels)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r2prof; mkdir -p $O
timeout 1500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/ps250 -o ps --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e --read-len 250 --sub 0.04 --indel 0.01 > $O/bench_250_pmcsq.json 2> $O/ps250.err
python3 $R/scripts/pmc_summary.py /tmp/ps250 $O/pmc_sq_se250.json > /dev/null
python3 - <<PY
import json
s=json.load(open("$O/pmc_sq_se250.json"))
for k in ("search_se_kernel","dp_kernel","finalize_se_kernel"):
    if k in s: print(k, {a: round(b["avg"]/1e6,1) for a,b in s[k].items()}, s[k]["SQ_INSTS_VALU"]["dispatches"])
PY
