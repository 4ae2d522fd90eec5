# GPU box: the default bench line twice -- index built in the run (as the driver runs it), then from the cache -- to see which of the two
# the file-to-file leg's slow runs belong to
mkdir -p gpurun_out/r5h
show() { python - <<PY
import json
d=json.loads(open('gpurun_out/r5h/$1.json').read().strip().splitlines()[-1])
e=d['e2e']
print('$1:', d['value'], d['ms_per_step'], 'e2e', round(e['value']/1e6,2), 'first', e['first_run_seconds'], 'sec', e['seconds'], e['stage_busy_s'], 'null', round(e['null_sink']['value']/1e6,2), e['null_sink']['lane_busy_s_summed'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', round(e['sharded']['value']/1e6,2), 'gz', {k: round(v['value']/1e6,2) for k,v in e.get('gz',{}).items()}, 'pairs', round(e.get('pairs',{}).get('value',0)/1e6,2))
PY
}
URMAPX_PIPE_TRACE=1 python bench.py > gpurun_out/r5h/full_uncached.json 2> gpurun_out/r5h/full_uncached.err; show full_uncached
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python bench.py --steps 2 --warmup 1 --no-other-workloads --no-cpu-baseline --no-e2e > /dev/null 2>&1
python bench.py > gpurun_out/r5h/full_cached.json 2> gpurun_out/r5h/full_cached.err; show full_cached
rm -rf /dev/shm/urmap_idx
grep -c . gpurun_out/r5h/full_uncached.err
