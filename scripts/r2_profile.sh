#!/bin/bash
# Round-2 profiles at hg38 scale (index cached in /dev/shm across the passes of this one call):
#  1 plain bench (all workloads)  2 rocprofv3 --kernel-trace --stats  3 --pmc FETCH_SIZE  4 --pmc SQ counters (two passes)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/r2prof; mkdir -p $O
T="timeout 1200"
A="--steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e"
$T python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_se.json 2> $O/bench_se.err; tail -c 300 $O/bench_se.err
$T rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py $A > $O/bench_se_ktrace.json 2> $O/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_se.csv
$T rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python3 $R/bench.py $A > $O/bench_se_pmcfetch.json 2> $O/pf.err
python3 $R/scripts/pmc_summary.py /tmp/pf $O/pmc_fetch_se_raw.json > /dev/null
$T rocprofv3 --pmc WRITE_SIZE -d /tmp/pw -o pw --output-format csv -- python3 $R/bench.py $A > $O/bench_se_pmcwrite.json 2> $O/pw.err
python3 $R/scripts/pmc_summary.py /tmp/pw $O/pmc_write_se_raw.json > /dev/null
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/ps -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_se_pmcsq.json 2> $O/ps.err
python3 $R/scripts/pmc_summary.py /tmp/ps $O/pmc_sq_se.json > /dev/null
$T rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES -d /tmp/ps2 -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_se_pmcsq2.json 2> $O/ps2.err
python3 $R/scripts/pmc_summary.py /tmp/ps2 $O/pmc_sq2_se.json > /dev/null
# 6: the file-to-file leg (text kernels of text_gpu.hip among the mapping kernels) under the kernel trace
URMAP_BENCH_NO_REFERENCE=1 $T rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/bench_e2e_ktrace.json 2> $O/kt2.err
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_e2e.csv
rm -rf /dev/shm/urmap_idx
ls -la $O | head -30
