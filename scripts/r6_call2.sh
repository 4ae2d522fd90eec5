# GPU box, round 6 call 2: the whole GPU suite on the prefetching kernels, then the same box alternating between the shipped library and
# the build without the L2 touches (URX_PREFETCH=0), the phase cycle shares of the diagnostic kernel, and kernel time against resident waves
mkdir -p gpurun_out/r6b
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6b/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r6b/pytest_gpu.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in pf nopf pf nopf; do
  if [ $v = pf ]; then unset URMAPX_LIB; else export URMAPX_LIB=$PWD/urmap_amd/csrc/build_nopf/liburmapx.so; fi
  python bench.py --no-e2e --no-cpu-baseline > gpurun_out/r6b/$v.json 2> gpurun_out/r6b/$v.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6b/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], [(n, o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
unset URMAPX_LIB
URMAPX_PHASE_STATS=1 python scripts/read_cost.py 3100 150 0.01 0.001 200000 > gpurun_out/r6b/read_cost_pf.txt 2>&1
head -8 gpurun_out/r6b/read_cost_pf.txt
URMAPX_LIB=$PWD/urmap_amd/csrc/build_nopf/liburmapx.so URMAPX_PHASE_STATS=1 python scripts/read_cost.py 3100 150 0.01 0.001 200000 > gpurun_out/r6b/read_cost_nopf.txt 2>&1
head -8 gpurun_out/r6b/read_cost_nopf.txt
python scripts/r6_waves_sweep.py 3100 1000000 > gpurun_out/r6b/waves_sweep.txt 2>&1
cat gpurun_out/r6b/waves_sweep.txt | tail -20
rm -rf /dev/shm/urmap_idx
