R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for st in 1 2 3; do
  timeout 1200 python3 bench.py --streams $st --steps 10 --warmup 2 --no-e2e --no-cpu-baseline > gpurun_out/r2/bench_streams$st.json 2> gpurun_out/r2/bench_streams$st.err; echo "streams $st rc=$?"
  python3 - $st <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/r2/bench_streams{sys.argv[1]}.json').read().strip().splitlines()[-1])
print(' ', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:12],k['avg_ms']) for k in d['kernels']])
for n,v in d.get('other_workloads',{}).items(): print('  ', n, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'])
PY
done
rm -rf /dev/shm/urmap_idx
