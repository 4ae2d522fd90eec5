# GPU tests then stage timings at 800 Mbp (150 and 250 bp) with parity on 100k reads
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r2/pytest_gpu.txt
bash scripts/r2_sweeps.sh
