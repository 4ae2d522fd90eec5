R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
export SWEEP_CHECK=${SWEEP_CHECK:-100000}
python3 scripts/stop_sweep.py ${MBP:-800} 150 0.01 0.001 1000000 ${STOPS:-0} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/sweep150.txt
python3 scripts/stop_sweep.py ${MBP:-800} 250 0.04 0.01 1000000 ${STOPS:-0} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/sweep250.txt
