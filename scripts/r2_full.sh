#!/bin/bash
# whole -m gpu suite, then the default bench (hg38 scale, all workloads, e2e) exactly as the driver runs it
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r2/pytest_gpu.txt
SECONDS=0; timeout 1500 python3 bench.py > gpurun_out/r2/bench_default.json 2> gpurun_out/r2/bench_default.err; echo "bench rc=$? wall ${SECONDS}s"

python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d.get('phase6'))
for k in d['kernels']: print(' ', k['kernel'][:40], k['avg_ms'], k['alg_bytes_per_read'], k['frac'], k.get('hbm_read_bytes_per_launch_pmc'))
print(d['roofline'])
print(d['cpu_baseline'])
for n,v in d.get('other_workloads',{}).items():
    print(n, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], [(k['kernel'][:20],k['avg_ms']) for k in v['kernels']])
print(d.get('e2e'))
print(d['config']['setup_s'])
PY
