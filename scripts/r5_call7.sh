#!/bin/bash
# GPU box: the text / pipeline tests on the deferred copy-back and the pooled lanes, then watched full bench runs
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5w; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_text.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_text.txt; cat $O/pytest_text.txt
NO_TRACE=1 bash scripts/r5_e2e_watch.sh
