R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
SWEEP_CHECK=200000 python3 scripts/stop_sweep.py 3100 150 0.01 0.001 1000000 0 2>&1 | grep "production\|parity"
SWEEP_CHECK=200000 python3 scripts/stop_sweep.py 3100 250 0.04 0.01 1000000 0 2>&1 | grep "production\|parity"
rm -rf /dev/shm/urmap_idx
