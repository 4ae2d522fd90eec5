# GPU box, round 6 call 36: the pair kernel's letter planes kept as one interleaved stream in LDS, slot_from_planes cutting a k-mer out of it (URX_SLOT_STREAM) against the library of the last full run (variants/r6d):
# the parity module, then A/B alternating, two rounds
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r7a; O=$R/gpurun_out/r7a
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullscale.py tests/test_gpu_pe_general.py tests/test_gpu_slow.py -q -m gpu -x 2>&1 | tail -5 ) > $O/tests.txt 2>&1
tail -2 $O/tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in r6d new r6d new; do
  if [ $v = new ]; then unset URMAPX_LIB; else export URMAPX_LIB=$R/urmap_amd/variants/$v/liburmapx.so; fi
  python bench.py --no-e2e --no-cpu-baseline > $O/$v.json 2> $O/$v.err
  python - <<PY
import json
d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['sequential']['value'], [(k['kernel'][:18],k['avg_ms']) for k in d['kernels'][:3]], [(n, o['value'], o['ms_per_step'], o['parity']['bit_identical_to_oracle'], o['sequential']['ms_per_step'], o['kernels'][0]['avg_ms']) for n,o in d['other_workloads'].items()])
PY
done
unset URMAPX_LIB
rm -rf /dev/shm/urmap_idx
