# GPU box, round 6 calls 16, 22 and 30: the code that ships (16 hardware queues): whole GPU suite + smoke(), the default bench line
mkdir -p gpurun_out/r6p
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6p/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r6p/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6p/smoke.txt 2>&1; tail -1 gpurun_out/r6p/smoke.txt
( time python bench.py ) > gpurun_out/r6p/bench_default.json 2> gpurun_out/r6p/bench_default.err
tail -3 gpurun_out/r6p/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6p/bench_default.json').read().strip().splitlines()[0])
c=d['config']
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['contexts'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['traffic'], c['inputs_are_the_recorded_ones'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], v['sequential']['value'])
e=d['e2e']
print('e2e', e['value'], e['sam_records_identical_to_oracle'], 'null', e['null_sink']['all_runs_reads_per_s'], 'sharded', e['sharded']['value'], e['sharded']['cat_of_shards_equals_the_one_file'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'], e['pairs']['sam_records_identical_to_oracle'], 'cli', e['cli']['index_streamed_to_the_device']['wall_s'], e['reference_binary']['reads_per_s'])
PY
