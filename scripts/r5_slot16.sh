# GPU box: the 16-byte slot table (DevIndex::slot16: position, tally, row length, second position / row index in one gather): parity, then A/B
mkdir -p gpurun_out/r5l
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullscale.py tests/test_gpu_slow.py tests/test_gpu_text.py -x -q -m gpu > gpurun_out/r5l/tests.txt 2>&1
tail -4 gpurun_out/r5l/tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in s16 rows s16 rows; do
  if [ $v = s16 ]; then unset URMAPX_NO_SLOT16; else export URMAPX_NO_SLOT16=1; fi
  python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r5l/$v.json 2> gpurun_out/r5l/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5l/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], d['config']['setup_s']['chain_row_bytes'], [(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
rm -rf /dev/shm/urmap_idx
