# hg38-scale paired-end and 250 bp benches sharing one index build (no CPU baseline); outputs under gpurun_out/final/
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/final; mkdir -p $O
timeout 200 python3 $R/bench.py --mode pe --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_pe_v11.json 2> $O/bench_pe_v11.err
timeout 40 python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se250_v11.json 2> $O/bench_se250_v11.err
rm -rf /dev/shm/urmap_idx
for f in bench_pe_v11 bench_se250_v11; do python3 -c "
import json
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], [(k['kernel'],k['avg_ms']) for k in d['kernels']], d['parity'])"; done
