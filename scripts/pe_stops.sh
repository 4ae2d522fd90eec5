#!/bin/bash
# diagnostic: PE search kernel time with the schedule cut after stage N (URMAPX_DEBUG_STOP_PE)
for stop in ${STOPS:-1 2 3 4 0}; do
  export URMAPX_DEBUG_STOP_PE=$stop
  echo "== stop $stop: $(timeout 300 python bench.py --mode pe --genome-mbp ${MBP:-800} --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"search_pe_kernel", "avg_ms": [0-9.]*')"
done
