# GPU box, round 6 call 6: the pair kernel with BOTH hit words in LDS at four waves per SIMD (URX_PE_DIET=2: 10 240 B) against the diet with the second
# word in global scratch (tail; default build), same box, alternating; the pair tests on the diet-2 library first
mkdir -p gpurun_out/r6f
( URMAPX_LIB=$PWD/urmap_amd/csrc/build_lds2/liburmapx.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_pe_general.py tests/test_gpu_slow.py tests/test_gpu_fullscale.py tests/test_gpu_text.py tests/test_gpu_multi.py -q -m gpu -k "pe or pair or Pair or PE or map2" 2>&1 | tail -8 ) > gpurun_out/r6f/lds2_tests.txt 2>&1
tail -3 gpurun_out/r6f/lds2_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in lds2 tail lds2 tail nopf; do
  if [ $v = tail ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/build_$v/liburmapx.so; fi
  URMAPX_VERBOSE=1 python bench.py --mode pe --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r6f/$v.json 2> gpurun_out/r6f/$v.err
  grep -m1 "search_pe_kernel grid" gpurun_out/r6f/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6f/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]])
PY
done
unset URMAPX_LIB
rm -rf /dev/shm/urmap_idx
