# GPU box, round 6 call 20: the .gz road with rounds written straight into the chunk buffers: the text tests, then the bench's e2e legs (gz included) twice
mkdir -p gpurun_out/r6r
( python -m pytest tests/test_gpu_text.py -q -m gpu -x 2>&1 | tail -3 ) > gpurun_out/r6r/text_tests.txt 2>&1
tail -2 gpurun_out/r6r/text_tests.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_E2E_PAIRS=1 URMAP_BENCH_NO_CLI=1 URMAP_BENCH_NO_REFERENCE=1
for v in a b; do
  URMAPX_PGZIP_VERBOSE=1 python bench.py --no-other-workloads --no-cpu-baseline > gpurun_out/r6r/e2e_$v.json 2> gpurun_out/r6r/e2e_$v.err
  grep "pgzip round" gpurun_out/r6r/e2e_$v.err | tail -4
  python - <<PY
import json
d=json.loads(open('gpurun_out/r6r/e2e_$v.json').read().strip().splitlines()[-1])
e=d['e2e']
print('$v', d['value'], 'e2e', e['value'], 'null', e['null_sink']['value'], 'gz', {k:(v['value'], v['inflate_GBs'], v['sam_records_identical_to_plain_run']) for k,v in e['gz'].items()})
PY
done
URMAP_BENCH_E2E_GZ_READS=10000000 python bench.py --no-other-workloads --no-cpu-baseline > gpurun_out/r6r/e2e_gz10m.json 2> gpurun_out/r6r/e2e_gz10m.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6r/e2e_gz10m.json').read().strip().splitlines()[-1])
e=d['e2e']
print('10 M reads gz', {k:(v['value'], v['inflate_GBs'], v['made_in_s'], v['sam_records_identical_to_plain_run']) for k,v in e['gz'].items()})
PY
rm -rf /dev/shm/urmap_idx
