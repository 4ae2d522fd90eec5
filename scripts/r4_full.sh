#!/bin/bash
# whole -m gpu suite, then the default bench (hg38 scale, all workloads, e2e incl. the .gz legs) exactly as the driver runs
# it, then the two-rank bench with its file-to-file leg over both ranks' devices
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4/full
if [ -z "$SKIP_TESTS" ]; then timeout 1800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4/full/pytest_gpu.txt 2>&1; tail -4 gpurun_out/r4/full/pytest_gpu.txt | cut -c1-300; fi
SECONDS=0; timeout 1500 python3 bench.py > gpurun_out/r4/full/bench_default.json 2> gpurun_out/r4/full/bench_default.err; echo "bench rc=$? wall ${SECONDS}s"
tail -c 300 gpurun_out/r4/full/bench_default.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/full/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d.get('phase6'))
for k in d['kernels']: print(' ', k['kernel'][:40], k['avg_ms'], k['alg_bytes_per_read'], k['frac'], k.get('hbm_read_bytes_per_launch_pmc'))
print(d['roofline'])
print(d['cpu_baseline'])
for n,v in d.get('other_workloads',{}).items():
    print(n, v['value'], v['ms_per_step'], v['parity']['bit_identical_to_oracle'], [(k['kernel'][:20],k['avg_ms']) for k in v['kernels']])
e=d.get('e2e',{})
print({k:v for k,v in e.items() if k not in ('gz','pairs','what')})
print(e.get('gz')); print(e.get('pairs'))
print(d['config']['setup_s'])
PY
if [ -z "$SKIP_RANKS" ]; then
SECONDS=0
URMAP_BENCH_E2E_READS=2000000 timeout 1500 python3 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r4/full/bench_2ranks.json 2> gpurun_out/r4/full/bench_2ranks.err; echo "2 ranks rc=$? wall ${SECONDS}s"
tail -c 300 gpurun_out/r4/full/bench_2ranks.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r4/full/bench_2ranks.json').read().splitlines() if l.startswith('{')][-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['config']['ranks'])
print({k:v for k,v in d.get('e2e',{}).items() if k not in ('what',)})
PY
fi
SECONDS=0; timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/full/bench_driver_cmd.json 2> gpurun_out/r4/full/bench_driver_cmd.err; echo "driver cmd rc=$? wall ${SECONDS}s"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/full/bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], d['roofline'])
PY
