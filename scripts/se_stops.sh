#!/bin/bash
# diagnostic: SE search kernel time with the schedule cut after step N (URMAPX_DEBUG_STOP), no profiler
for stop in ${STOPS:-1 3 104 401 402 403 4 0}; do
  export URMAPX_DEBUG_STOP=$stop
  echo "== stop $stop: $(timeout 400 python bench.py --genome-mbp ${MBP:-800} ${ARGS} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"search_se_kernel", "avg_ms": [0-9.]*')"
done
