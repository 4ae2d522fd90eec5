# GPU box: phase 6's round boundaries (URMAPX_DP_BOUNDS), 150 and 250 bases, on the final kernels
mkdir -p gpurun_out/r5n
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for b in default 0,16 0,4 0,4,32 0,1,4,16; do
  if [ $b = default ]; then unset URMAPX_DP_BOUNDS; else export URMAPX_DP_BOUNDS=$b; fi
  python bench.py --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r5n/se150_$b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5n/se150_$b.json').read().strip().splitlines()[-1])
print('150 bp bounds $b:', d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:10],k['avg_ms']) for k in d['kernels'][:3]], d['phase6']['launch_ms_by_round']['dp_kernel'], d['phase6']['dropped_by_a_round_gate_before_their_dp'])
PY
done
for b in default 0,4,32 0,2,16,64 0,3,12,48 0,2,6,24; do
  if [ $b = default ]; then unset URMAPX_DP_BOUNDS; else export URMAPX_DP_BOUNDS=$b; fi
  python bench.py --no-e2e --no-cpu-baseline --no-other-workloads --read-len 250 --sub 0.04 --indel 0.01 --steps 6 > gpurun_out/r5n/se250_$b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5n/se250_$b.json').read().strip().splitlines()[-1])
print('250 bp bounds $b:', d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:10],k['avg_ms']) for k in d['kernels'][:3]], d['phase6']['launch_ms_by_round']['dp_kernel'], d['phase6']['dropped_by_a_round_gate_before_their_dp'])
PY
done
rm -rf /dev/shm/urmap_idx
