# GPU box: the default bench line three times on one box -- is the file-to-file leg steady now that idle OpenMP workers sleep?
mkdir -p gpurun_out/r5i
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for i in 1 2 3; do
  python bench.py > gpurun_out/r5i/run$i.json 2> gpurun_out/r5i/run$i.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5i/run$i.json').read().strip().splitlines()[-1])
e=d['e2e']; n=e['null_sink']
print('run $i:', d['value'], d['ms_per_step'], 'e2e', round(e['value']/1e6,2), e['seconds'], 'null', round(n['value']/1e6,2), n['seconds'], n['lane_busy_s_summed'], n['stream_time_s_summed_over_lanes'], 'sharded', round(e['sharded']['value']/1e6,2), 'gz', {k: round(v['value']/1e6,2) for k,v in e['gz'].items()}, 'pairs', round(e['pairs']['value']/1e6,2), e['placement'])
PY
done
rm -rf /dev/shm/urmap_idx
