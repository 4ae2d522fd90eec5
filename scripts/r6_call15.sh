# GPU box, round 6 call 15: is the bench line's null-sink leg slowed by HIP streams sharing hardware queues (GPU_MAX_HW_QUEUES, default 4)?
mkdir -p gpurun_out/r6o
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1 URMAP_BENCH_NO_CLI=1 URMAP_BENCH_NO_REFERENCE=1
for q in default 8 16 default; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --no-other-workloads --no-cpu-baseline > gpurun_out/r6o/q_$q.json 2> gpurun_out/r6o/q_$q.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6o/q_$q.json').read().strip().splitlines()[-1])
e=d['e2e']
print('GPU_MAX_HW_QUEUES $q:', d['value'], d['ms_per_step'], d['sequential']['value'], 'e2e', e['value'], 'null', e['null_sink']['all_runs_reads_per_s'], 'sharded', e['sharded']['value'], e['null_sink']['lanes_view']['stream_time_s'])
PY
done
rm -rf /dev/shm/urmap_idx
