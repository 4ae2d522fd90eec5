# GPU box, round 6 call 24: (1) which instruction-cache counters the box offers; (2) A/B base against the one-site scan-ahead loop (ahead3), two rounds;
# (3) instruction-fetch counters of the single-end search kernel for both (counters in runs of their own, one context)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6r; O=$R/gpurun_out/r6r
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
rocprofv3 -L 2>/dev/null | grep -iE "icache|ifetch|SQC_|WAIT_INST|INST_LEVEL|INST_CYCLES" | cut -c1-200 | head -80 > $O/counters_list.txt
wc -l $O/counters_list.txt
for v in base ahead3 base ahead3; do
  export URMAPX_LIB=$R/urmap_amd/variants/$v/liburmapx.so
  python bench.py --no-e2e --no-cpu-baseline > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['sequential']['value'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], [(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['sequential']['ms_per_step'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
cd /tmp; export TMPDIR=/tmp
A="--steps 4 --warmup 1 --contexts 1 --no-cpu-baseline --no-other-workloads --no-e2e"
for v in base ahead3; do
  export URMAPX_LIB=$R/urmap_amd/variants/$v/liburmapx.so
  i=0
  for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAVES" "SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $C -d /tmp/pc_${v}_$i -o pc --output-format csv -- python3 $R/bench.py $A > $O/bench_${v}_pmc$i.json 2> $O/pmc_${v}_$i.err
    python3 $R/scripts/pmc_summary.py /tmp/pc_${v}_$i $O/pmc_${v}_$i.json > /dev/null 2>&1
    python3 - <<PY
import json
try:
    d=json.load(open("$O/pmc_${v}_$i.json"))
    for k,x in d.items():
        if k=='search_se_kernel': print("$v", {c: round(y['avg']/1e6,2) for c,y in x.items()})
except Exception as e: print("$v $i failed", e)
PY
    rm -rf /tmp/pc_${v}_$i
  done
done
rm -rf /dev/shm/urmap_idx
