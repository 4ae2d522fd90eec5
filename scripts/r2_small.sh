R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
URMAP_BENCH_E2E_READS=${E2E:-1000000} timeout 900 python3 bench.py --genome-mbp ${MBP:-200} --reads-per-step 500000 --steps 3 --warmup 1 > gpurun_out/r2/bench_small.json 2> gpurun_out/r2/bench_small.err; echo rc=$?; tail -c 600 gpurun_out/r2/bench_small.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_small.json').read().strip().splitlines()[-1])
print(d['value'], d['parity']['bit_identical_to_oracle']); print(json.dumps(d.get('e2e'),indent=1)); print(json.dumps(d.get('cpu_baseline'),indent=1))
for n,v in d.get('other_workloads',{}).items(): print(n, v['value'], v['parity']['bit_identical_to_oracle'])
PY
