# default bench (hg38 scale, with the CPU baseline) + rocprofv3 kernel trace of the same command; outputs under gpurun_out/final/
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/final; mkdir -p $O
timeout 600 python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_se.json 2> $O/bench_se.err
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_se_ktrace.json 2> $O/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_se.csv
rm -rf /dev/shm/urmap_idx
python3 -c "
import json
d=json.loads(open('$O/bench_se.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']], d['parity'], d['cpu_baseline']['value'], d['roofline'])"
head -5 $O/kernel_stats_se.csv
