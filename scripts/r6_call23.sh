# GPU box, round 6 call 23: the candidate scan split in two (URX_SCAN_AHEAD): single-end tests on the new library, then A/B on one box, alternating libraries --
# base (the loop as shipped so far), ahead1 (scan of the next batch issued in front of this batch's window loads), ahead2 (behind them, windows held in registers)
mkdir -p gpurun_out/r6q
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_slow.py tests/test_gpu_fullscale.py tests/test_gpu_phase3.py -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r6q/ahead1_tests.txt 2>&1
tail -3 gpurun_out/r6q/ahead1_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in base ahead1 ahead2 base ahead1 ahead2; do
  if [ $v = ahead1 ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/variants/$v/liburmapx.so; fi
  python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r6q/$v.json 2> gpurun_out/r6q/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6q/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['sequential']['value'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], [(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['sequential']['ms_per_step'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
unset URMAPX_LIB
rm -rf /dev/shm/urmap_idx
