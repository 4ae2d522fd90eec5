# GPU box, round 6 call 9: slot16 + rows built in one go (no per-slot info entries in between) against the two-step build of round 5: the layout tests,
# the full-scale module, and the headline bench on both builds with the layouts' checksums
mkdir -p gpurun_out/r6i
( python -m pytest tests/test_gpu_parity.py tests/test_gpu_validate.py tests/test_gpu_fullscale.py tests/test_gpu_bigfiles.py -q -m gpu -k "chain_rows or dense or validate or fullscale or oracle or checksum or recorded" 2>&1 | tail -6 ) > gpurun_out/r6i/tests.txt 2>&1
tail -3 gpurun_out/r6i/tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for v in direct twostep direct; do
  if [ $v = direct ]; then unset URMAPX_TWO_STEP_LAYOUT; else export URMAPX_TWO_STEP_LAYOUT=1; fi
  URMAPX_VERBOSE=1 python bench.py --no-e2e --no-cpu-baseline --no-other-workloads > gpurun_out/r6i/$v.json 2> gpurun_out/r6i/$v.err
  grep -E "slot16|chain rows" gpurun_out/r6i/$v.err | head -3
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6i/$v.json').read().strip().splitlines()[-1])
c=d['config']
print('$v', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], c['slot16_checksum'], c['chain_rows_checksum'], c['slot_table_checksum'], c['setup_s'], c['ranks']['index_bytes_per_rank'])
PY
done
rm -rf /dev/shm/urmap_idx
