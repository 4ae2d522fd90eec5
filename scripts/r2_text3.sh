#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/text3
timeout 600 python -m pytest tests/test_gpu_text.py -x -q -m gpu 2>&1 | tail -3
URMAPX_PIPE_TRACE=1 timeout 1200 python scripts/e2e_probe.py --genome-mbp 400 --reads ${READS:-4000000} --repeat 3 \
  --set text:2:262144 --set text:2:262144:URMAPX_WRITE_THREADS=2 --set text:2:262144:URMAPX_WRITE_THREADS=4 --set text:3:262144 --set text:2:393216 \
  > gpurun_out/text3/probe.log 2> gpurun_out/text3/probe.err
cat gpurun_out/text3/probe.log
grep -v "^trace" gpurun_out/text3/probe.err | tail -3
