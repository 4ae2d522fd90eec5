R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
timeout 1500 python3 bench.py --steps 10 --warmup 2 > gpurun_out/r2/bench_se.json 2> gpurun_out/r2/bench_se.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r2/bench_se.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_se.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity'], d.get('phase6'))
for k in d['kernels']: print(' ', k['kernel'][:40], k['avg_ms'], k['alg_bytes_per_read'], k['frac'])
print(d['cpu_baseline']['value'], d['cpu_baseline'].get('value_10_threads'))
for n,v in d.get('other_workloads',{}).items():
    print(n, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], [(k['kernel'][:20],k['avg_ms']) for k in v['kernels']])
print(d['work_per_read'])
PY
