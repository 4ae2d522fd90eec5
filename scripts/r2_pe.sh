R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r2
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "pe or map2 or pair" 2>&1 | tail -4
MBP=${MBP:-3100} bash scripts/r2_pe_trace.sh
