# GPU box, round 6 call 25: search_se_kernel<3, ..., KCH = 2> (two chunks of k-mer starts for reads of up to 151 bases at W = 24, row store in LDS):
# the single-end / full-scale / text tests, then A/B in ONE library (URMAPX_NO_K2=1 launches the three-chunk instance as before), alternating, two rounds
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r6s; O=$R/gpurun_out/r6s
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_slow.py tests/test_gpu_fullscale.py tests/test_gpu_phase3.py tests/test_gpu_text.py -q -m gpu -x 2>&1 | tail -8 ) > $O/k2_tests.txt 2>&1
tail -3 $O/k2_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in k3 k2 k3 k2; do
  if [ $v = k3 ]; then export URMAPX_NO_K2=1; else unset URMAPX_NO_K2; fi
  python bench.py --no-e2e --no-cpu-baseline --no-other-workloads > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['sequential']['value'], d['sequential']['ms_per_step'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]])
PY
done
unset URMAPX_NO_K2
rm -rf /dev/shm/urmap_idx
