# GPU box: what the driver runs at round end (pytest -m gpu, smoke), the parity fuzz at 800 Mbp, the default bench line
mkdir -p gpurun_out/r5z
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r5z/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r5z/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5z/smoke.txt 2>&1; tail -2 gpurun_out/r5z/smoke.txt
python scripts/fuzz_gpu.py 800 200000 > gpurun_out/r5z/fuzz_800mbp.txt 2>&1; tail -18 gpurun_out/r5z/fuzz_800mbp.txt
( time python bench.py ) > gpurun_out/r5z/bench_default.json 2> gpurun_out/r5z/bench_default.err
tail -4 gpurun_out/r5z/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5z/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['write_bytes'], d['config']['index_validated'])
e=d['e2e']
print('e2e', e['value'], 'null', e['null_sink']['value'], 'sharded', e['sharded']['value'], 'gz', {k:v['value'] for k,v in e['gz'].items()}, 'pairs', e['pairs']['value'], e['placement'])
print(d['probe_only'])
PY
# the kernel timeline of the file-to-file legs on the same code (scripts/e2e_timeline.py)
if [ -z "$NO_TRACE" ]; then E2E_READS=4000000 bash scripts/r5_e2e_trace.sh; fi
