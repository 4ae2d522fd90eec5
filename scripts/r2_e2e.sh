R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cli or map_files or two_devices" 2>&1 | tail -4
URMAP_BENCH_E2E_READS=8000000 timeout 900 python3 bench.py --genome-mbp 800 --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r2/bench_e2e.json 2> gpurun_out/r2/bench_e2e.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_e2e.json').read().strip().splitlines()[-1])
print(d['value'], json.dumps(d.get('e2e'),indent=1))
PY
