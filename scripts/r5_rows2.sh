# GPU box: the row layout with the second position inline (chain_rows.hip, rows_fetch, the pair kernel's lookup): parity, then the three device workloads
mkdir -p gpurun_out/r5j
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullscale.py tests/test_gpu_phase3.py tests/test_gpu_slow.py tests/test_gpu_pe_general.py -x -q -m gpu > gpurun_out/r5j/tests.txt 2>&1
tail -4 gpurun_out/r5j/tests.txt
python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r5j/bench.json 2> gpurun_out/r5j/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5j/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], d['config']['setup_s'])
for n,o in d['other_workloads'].items(): print(n, o['value'], o['ms_per_step'], o['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in o['kernels'][:3]])
PY
