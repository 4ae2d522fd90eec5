# GPU box, round 6 call 5: the pair kernel on its diet with hits 65..128 of a mate in global scratch (tail; default build) against the diet whose
# first pass flags such pairs for the second pass (notail), same box, alternating; the pair tests on the tail build first
mkdir -p gpurun_out/r6e
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_pe_general.py tests/test_gpu_slow.py tests/test_gpu_fullscale.py tests/test_gpu_text.py tests/test_gpu_multi.py -q -m gpu -k "pe or pair or Pair or PE or map2" 2>&1 | tail -8 ) > gpurun_out/r6e/tail_tests.txt 2>&1
tail -3 gpurun_out/r6e/tail_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in tail notail tail notail; do
  if [ $v = tail ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/build_$v/liburmapx.so; fi
  python bench.py --mode pe --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r6e/$v.json 2> gpurun_out/r6e/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6e/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]])
PY
done
unset URMAPX_LIB
rm -rf /dev/shm/urmap_idx
