R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/final; mkdir -p $O
timeout 900 python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_se.json 2> $O/bench_se.err
timeout 900 python3 $R/bench.py --mode pe --steps 5 --warmup 1 > $O/bench_pe.json 2> $O/bench_pe.err
rm -rf /dev/shm/urmap_idx
for f in bench_se bench_pe; do python3 -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']], d['parity'], d['cpu_baseline']['value'])"; done
