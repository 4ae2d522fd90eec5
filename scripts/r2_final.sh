#!/bin/bash
# the driver's command line, then the GPU suite
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
SECONDS=0; timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2/bench_final.json 2> gpurun_out/r2/bench_final.err; echo "bench rc=$? wall ${SECONDS}s"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity'], d.get('phase6'))
for k in d['kernels']: print(' ', k['kernel'][:40], k['avg_ms'], k['alg_bytes_per_read'], k['frac'], k.get('hbm_read_bytes_per_launch_pmc'))
print(d['roofline'])
print({k:v for k,v in d['cpu_baseline'].items() if k!='sample'})
for n,v in d.get('other_workloads',{}).items():
    print(n, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], [(k['kernel'][:20],k['avg_ms']) for k in v['kernels']], v['cpu_port_reads_per_s'])
print(d.get('e2e'))
print(d['config']['setup_s'])
PY
timeout 1800 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
