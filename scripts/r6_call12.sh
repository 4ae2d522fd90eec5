# GPU box, round 6 call 12: the code that ships after the two-context bench and the one-go layouts: whole GPU suite + smoke(), the default bench line,
# and the kernel-only line with one and with three contexts for comparison
mkdir -p gpurun_out/r6l
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6l/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r6l/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6l/smoke.txt 2>&1; tail -1 gpurun_out/r6l/smoke.txt
( time python bench.py ) > gpurun_out/r6l/bench_default.json 2> gpurun_out/r6l/bench_default.err
tail -3 gpurun_out/r6l/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6l/bench_default.json').read().strip().splitlines()[0])
c=d['config']
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['contexts'], d['sequential']['value'], d['sequential']['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['traffic'], c['inputs_are_the_recorded_ones'], c['slot16_checksum'])
for k,v in d['other_workloads'].items(): print(k, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], v['sequential']['value'])
e=d['e2e']
print('e2e', e['value'], e['sam_records_identical_to_oracle'], 'null', e['null_sink']['all_runs_reads_per_s'], 'sharded', e['sharded']['value'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'], e['pairs']['sam_records_identical_to_oracle'], 'cli', e['cli']['index_streamed_to_the_device']['wall_s'])
print([ (k['kernel'][:20], k['avg_ms'], k.get('timed_region_event_ms')) for k in d['kernels']], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for c in 1 3 2; do
  python bench.py --contexts $c --no-e2e --no-cpu-baseline > gpurun_out/r6l/contexts_$c.json 2> gpurun_out/r6l/contexts_$c.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6l/contexts_$c.json').read().strip().splitlines()[-1])
print('contexts $c', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(n, o['value'], o['ms_per_step'], o['parity']['bit_identical_to_oracle']) for n,o in d['other_workloads'].items()])
PY
done
rm -rf /dev/shm/urmap_idx
