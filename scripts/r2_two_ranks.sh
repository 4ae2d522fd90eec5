# hg38-scale bench with two ranks on the one GPU of the box (gloo; index reaches rank 1 through the broadcast branch)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
SECONDS=0
URMAP_BENCH_BROADCAST=1 timeout 1500 python3 bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r2/bench_2ranks.json 2> gpurun_out/r2/bench_2ranks.err; echo "rc=$? wall ${SECONDS}s"
tail -c 500 gpurun_out/r2/bench_2ranks.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r2/bench_2ranks.json').read().splitlines() if l.startswith('{')][-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['parity'], d['config']['ranks'], d['config']['setup_s'])
PY
