#!/bin/bash
# quick look at the single-end step at hg38 scale: bench.py without the CPU baseline / other workloads / e2e
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
TAG=${TAG:-se}
timeout 900 python3 bench.py --no-cpu-baseline --no-other-workloads --no-e2e ${EXTRA} > gpurun_out/r3/bench_$TAG.json 2> gpurun_out/r3/bench_$TAG.err; echo rc=$?; tail -c 400 gpurun_out/r3/bench_$TAG.err
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r3/bench_$TAG.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['parity'].get('mismatches'), d.get('phase6'))
for k in d['kernels']: print(' ', k['kernel'][:40], k['avg_ms'], k['alg_bytes_per_read'], k['frac'])
print(d['config']['setup_s'])
PY
