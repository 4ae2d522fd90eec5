R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 900 python3 -m pytest tests -m gpu -x -q -k "make_ufi" > gpurun_out/r2/pytest_ufi.txt 2>&1; tail -8 gpurun_out/r2/pytest_ufi.txt
export URMAPX_VERBOSE=1
URMAP_BENCH_E2E_READS=8000000 timeout 1500 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r2/bench_se.json 2> gpurun_out/r2/bench_se.err; echo "bench rc=$?"
grep "make_ufi" gpurun_out/r2/bench_se.err | head -20
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_se.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['config']['setup_s'])
print(json.dumps(d.get('e2e'),indent=1))
for n,v in d.get('other_workloads',{}).items(): print(n, v['value'], v['parity']['bit_identical_to_oracle'])
PY
