# GPU box: the shipped library against a build with 5 waves per SIMD for the 150-base search kernel (96 VGPRs, HSP list in LDS cut to 64: 8.5 KB per block)
mkdir -p gpurun_out/r5k
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in base w5 base w5; do
  if [ $v = base ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/build_w5/liburmapx.so; fi
  python bench.py --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r5k/$v.json 2> gpurun_out/r5k/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5k/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]])
PY
done
rm -rf /dev/shm/urmap_idx
