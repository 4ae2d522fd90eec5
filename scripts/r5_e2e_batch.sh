# GPU box: the file-to-file leg (10 M reads) by chunk size: the library's choice (0) against fixed -batch values
mkdir -p gpurun_out/r5e
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
for b in ${BATCHES:-0 262144 524288 1048576}; do
  URMAP_BENCH_E2E_BATCH=$b python bench.py --steps 2 --warmup 1 --no-other-workloads --no-cpu-baseline > gpurun_out/r5e/bench_b$b.json 2> gpurun_out/r5e/bench_b$b.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5e/bench_b$b.json').read().strip().splitlines()[-1])
e=d['e2e']
print('batch $b: e2e', round(e['value']/1e6,2), 'first', e['first_run_seconds'], 'sec', e['seconds'], e['stage_busy_s'], 'null', round(e['null_sink']['value']/1e6,2), e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', round(e['sharded']['value']/1e6,2))
PY
done
rm -rf /dev/shm/urmap_idx
