# GPU box, round 6 call 7: the code that ships -- the whole GPU suite and smoke(), the lanes with the chunk ramp into files with two and three lanes, kernel time against
# resident waves on the shipped kernels (the floor model's data), the default bench line
mkdir -p gpurun_out/r6g
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6g/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r6g/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6g/smoke.txt 2>&1; tail -1 gpurun_out/r6g/smoke.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python scripts/r6_waves_sweep.py 3100 1000000 > gpurun_out/r6g/waves_sweep.txt 2>&1
tail -16 gpurun_out/r6g/waves_sweep.txt
python scripts/r6_lanes.py 3100 10000000 files > gpurun_out/r6g/lanes_files.txt 2>&1
grep -E "streams|one file|two shards" gpurun_out/r6g/lanes_files.txt
rm -rf /dev/shm/urmap_idx
unset URMAP_BENCH_INDEX_CACHE
( time python bench.py ) > gpurun_out/r6g/bench_default.json 2> gpurun_out/r6g/bench_default.err
tail -3 gpurun_out/r6g/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6g/bench_default.json').read().strip().splitlines()[0])
c=d['config']
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['genome_checksum'], c['slot_table_checksum'], c['inputs_are_the_recorded_ones'], c['index_validation']['used_slots'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'])
e=d['e2e']
print('e2e', e['value'], e['sam_slices_checked'], 'null', e['null_sink']['value'], 'sharded', e['sharded']['value'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'], e['pairs']['sam_slices_checked'], 'cli', e.get('cli',{}).get('index_streamed_to_the_device'))
print([ (k['kernel'][:20], k['avg_ms']) for k in d['kernels']], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
