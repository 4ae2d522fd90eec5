# GPU box, round 6 call 3: A/B of the search-kernel variants on one box, alternating libraries -- base (round-5 kernels), L2 touches for a read's long
# rows only, for the next batch's windows only, and the pair kernel on its LDS diet at four waves per SIMD -- after the pair tests on that library;
# then the lanes of urmapx_map_files against the blocks per CU the persistent search kernel takes
mkdir -p gpurun_out/r6c
( URMAPX_LIB=$PWD/urmap_amd/csrc/build_pediet/liburmapx.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_pe_general.py tests/test_gpu_slow.py tests/test_gpu_fullscale.py -q -m gpu -k "pe or pair or Pair or PE" 2>&1 | tail -8 ) > gpurun_out/r6c/pediet_tests.txt 2>&1
tail -3 gpurun_out/r6c/pediet_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in base pfrows pfwin pediet base pfrows pediet; do
  if [ $v = base ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/build_$v/liburmapx.so; fi
  python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r6c/$v.json 2> gpurun_out/r6c/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6c/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], [(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
unset URMAPX_LIB
python scripts/r6_lanes.py 3100 10000000 > gpurun_out/r6c/lanes_blocks.txt 2>&1
grep streams gpurun_out/r6c/lanes_blocks.txt
rm -rf /dev/shm/urmap_idx
