# GPU box: phase 3 parked -- its own tests, the parity suites that go through the single-end path, the full-scale module, a short bench
mkdir -p gpurun_out/r5c
python -m pytest tests/test_gpu_phase3.py -x -q -m gpu > gpurun_out/r5c/phase3_tests.txt 2>&1
tail -5 gpurun_out/r5c/phase3_tests.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_slow.py tests/test_gpu_text.py tests/test_gpu_validate.py -x -q -m gpu > gpurun_out/r5c/parity_tests.txt 2>&1
tail -5 gpurun_out/r5c/parity_tests.txt
python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu > gpurun_out/r5c/fullscale_tests.txt 2>&1
tail -3 gpurun_out/r5c/fullscale_tests.txt
python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r5c/bench_noe2e.json 2> gpurun_out/r5c/bench_noe2e.err
tail -c 400 gpurun_out/r5c/bench_noe2e.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench_noe2e.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('phase3'), d['parity']['bit_identical_to_oracle'])
for k in d['kernels']: print(k['kernel'][:40], k['avg_ms'], k.get('launches'))
print(d.get('probe_only'))
for n,o in d['other_workloads'].items(): print(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o.get('phase3'), [(k['kernel'][:20],k['avg_ms']) for k in o['kernels']])
PY
