# GPU box: the pair kernel's probe through slot16: the pair tests, then the device workloads
mkdir -p gpurun_out/r5o
python -m pytest tests/test_gpu_parity.py tests/test_gpu_pe_general.py tests/test_gpu_text.py tests/test_gpu_multi.py -x -q -m gpu -k "pe or pair or map2 or PE" > gpurun_out/r5o/tests.txt 2>&1
tail -3 gpurun_out/r5o/tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for i in 1 2; do
python bench.py --no-e2e --no-cpu-baseline --mode pe > gpurun_out/r5o/pe$i.json 2> gpurun_out/r5o/pe$i.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r5o/pe$i.json').read().strip().splitlines()[-1])
print('pe', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:2]])
PY
done
rm -rf /dev/shm/urmap_idx
