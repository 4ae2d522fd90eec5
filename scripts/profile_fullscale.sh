#!/bin/bash
# Round-end measurement on the GPU box: hg38-scale benches (SE with CPU baseline, PE, 250 bp) + rocprofv3 kernel trace
# and PMC passes of the SE bench.  Writes summaries under gpurun_out/full/ (copy into profiles/rN/ afterwards).
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/full; mkdir -p $O
T="timeout 900"
$T python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_se.json 2> $O/bench_se.err; tail -c 600 $O/bench_se.err
$T rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_se_ktrace.json 2> $O/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_se.csv
$T rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o pf --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se_pmcfetch.json 2> $O/pf.err
python3 $R/scripts/pmc_summary.py /tmp/pf $O/pmc_fetch_se_raw.json > /dev/null
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/ps -o ps --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se_pmcsq.json 2> $O/ps.err
python3 $R/scripts/pmc_summary.py /tmp/ps $O/pmc_sq_se.json > /dev/null
$T rocprofv3 --pmc FETCH_SIZE -d /tmp/pfpe -o pf --output-format csv -- python3 $R/bench.py --mode pe --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pe_pmcfetch.json 2> $O/pfpe.err
python3 $R/scripts/pmc_summary.py /tmp/pfpe $O/pmc_fetch_pe_raw.json > /dev/null
$T rocprofv3 --pmc FETCH_SIZE -d /tmp/pf250 -o pf --output-format csv -- python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se250_pmcfetch.json 2> $O/pf250.err
python3 $R/scripts/pmc_summary.py /tmp/pf250 $O/pmc_fetch_se250_raw.json > /dev/null
$T python3 $R/bench.py --mode pe --steps 5 --warmup 1 > $O/bench_pe.json 2> $O/bench_pe.err
$T rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt --output-format csv -- python3 $R/bench.py --mode pe --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pe_ktrace.json 2> $O/ktp.err
cp $(find /tmp/ktp -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pe.csv
$T python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 5 --warmup 1 > $O/bench_se250.json 2> $O/bench_se250.err
rm -rf /dev/shm/urmap_idx
ls -la $O; tail -c 300 $O/bench_se.json
