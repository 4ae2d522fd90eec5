#!/bin/bash
# SQ + instruction-cache counters of the single-end step at hg38 scale (index cached in /dev/shm across the passes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
TAG=${TAG:-v1}
O=$R/gpurun_out/r3/sq_$TAG; mkdir -p $O
T="timeout 900"
A="--steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e"
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/ps -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_pmcsq.json 2> $O/ps.err
python3 $R/scripts/pmc_summary.py /tmp/ps $O/pmc_sq.json > /dev/null
$T rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES -d /tmp/ps2 -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_pmcsq2.json 2> $O/ps2.err
python3 $R/scripts/pmc_summary.py /tmp/ps2 $O/pmc_sq2.json > /dev/null
$T rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT -d /tmp/pi -o pi --output-format csv -- python3 $R/bench.py $A > $O/bench_pmcic.json 2> $O/pi.err
python3 $R/scripts/pmc_summary.py /tmp/pi $O/pmc_icache.json > /dev/null
rm -rf /dev/shm/urmap_idx
python3 - <<PY
import json
for f in ("pmc_sq.json","pmc_sq2.json","pmc_icache.json"):
    try:
        d=json.load(open("$O/"+f))
    except Exception as e:
        print(f, e); continue
    for k,v in d.items():
        print(f, k, {c: round(x['avg']/1e6,2) for c,x in v.items()})
PY
tail -3 $O/pi.err
