R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 scripts/pe_sweep.py 3100 1000000 2>&1 | grep stop
