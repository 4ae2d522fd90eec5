# GPU box: slot16 also in the pair kernel's pending stage and in the 512 / 1024-base classes; info entries dropped after the build: suite, then bench
mkdir -p gpurun_out/r5m
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r5m/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r5m/pytest_gpu.txt
python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r5m/bench.json 2> gpurun_out/r5m/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5m/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], d['config']['setup_s'])
for n,o in d['other_workloads'].items(): print(n, o['value'], o['ms_per_step'], o['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in o['kernels'][:3]])
PY
