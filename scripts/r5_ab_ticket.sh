# GPU box: the shipped library (4 reads per ticket in the single-end search kernel) against builds with 2 and 8 (make EXTRA=-DURX_TICKET_CHUNK=n)
mkdir -p gpurun_out/r5tc
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in base tc2 tc8 base tc2 tc8; do
  if [ $v = base ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/ab/liburmapx_$v.so; fi
  python bench.py --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r5tc/$v.json 2> gpurun_out/r5tc/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5tc/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]])
PY
done
rm -rf /dev/shm/urmap_idx
