#!/bin/bash
# round 2: text path (FASTQ bytes -> SAM bytes on the device): tests, write-strategy microbenchmark, file-to-file probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/text
timeout 900 python -m pytest tests/test_gpu_text.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/text/pytest_text.log 2>&1
echo "pytest text rc=$?" > gpurun_out/text/rc.txt
tail -5 gpurun_out/text/pytest_text.log
g++ -O2 -fopenmp -o /tmp/fs_bench scripts/fs_bench.cpp 2> /dev/null
for m in "0 1" "1 8" "1 16" "2 4" "2 8" "2 16"; do /tmp/fs_bench /dev/shm/fsbench.tmp $m; done > gpurun_out/text/fs_bench.txt 2>&1
cat gpurun_out/text/fs_bench.txt
timeout 1200 python scripts/e2e_probe.py --genome-mbp 400 --reads 4000000 \
  --set host:2:262144 --set text:2:262144 --set text:2:262144:URMAPX_SAM_WRITE=pwrite --set text:3:131072 --set text:2:524288 --set text:1:262144 \
  > gpurun_out/text/probe.log 2> gpurun_out/text/probe.err
echo "probe rc=$?" >> gpurun_out/text/rc.txt
cat gpurun_out/text/probe.log; tail -5 gpurun_out/text/probe.err
