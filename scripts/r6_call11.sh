# GPU box, round 6 call 11: the bench line with whole batches alternating over two contexts (the new default): the multi-rank tests, the default line, and the
# kernel trace (rocprofv3 --kernel-trace --stats) of the three device workloads in that mode
mkdir -p gpurun_out/r6k
( python -m pytest tests/test_gpu_multi.py tests/test_gpu_fullscale.py -q -m gpu 2>&1 | tail -5 ) > gpurun_out/r6k/tests.txt 2>&1
tail -3 gpurun_out/r6k/tests.txt
( time python bench.py ) > gpurun_out/r6k/bench_default.json 2> gpurun_out/r6k/bench_default.err
tail -3 gpurun_out/r6k/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6k/bench_default.json').read().strip().splitlines()[0])
c=d['config']
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['contexts'], d['sequential'], d['roofline']['frac'], d['roofline']['kernel_alone'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], v['sequential'])
e=d['e2e']
print('e2e', e['value'], 'null', e['null_sink']['value'], e['null_sink']['all_runs_reads_per_s'], 'sharded', e['sharded']['value'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'])
print([ (k['kernel'][:20], k['avg_ms'], k.get('alone_ms')) for k in d['kernels']], d['cpu_baseline']['value'])
PY
R=$PWD; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for wl in se150 pe se250; do
  case $wl in
    se150) W="";;
    pe) W="--mode pe";;
    se250) W="--read-len 250 --sub 0.04 --indel 0.01";;
  esac
  A="--steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads --no-e2e $W"
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt2_$wl -o kt --output-format csv -- python3 $R/bench.py $A > $R/gpurun_out/r6k/bench_${wl}_ktrace.json 2> $R/gpurun_out/r6k/kt_$wl.err
  cp $(find /tmp/kt2_$wl -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6k/kernel_stats_${wl}_two_contexts.csv
  echo "== $wl"; grep -E "search_se_kernel<[34], false, false, 2|search_pe_kernel<3, 0|dp_kernel<[34]>" $R/gpurun_out/r6k/kernel_stats_${wl}_two_contexts.csv | cut -c1-60,200-330 | head -4
  python3 - <<PY
import json
d=json.loads(open('$R/gpurun_out/r6k/bench_${wl}_ktrace.json').read().strip().splitlines()[-1])
print('$wl under trace', d['value'], d['ms_per_step'], [(k['kernel'][:18], k['avg_ms'], k.get('alone_ms')) for k in d['kernels'][:3]])
PY
done
rm -rf /dev/shm/urmap_idx
