#!/bin/bash
# diagnostic: SQ counters of the search kernel with the schedule cut after step N (URMAPX_DEBUG_STOP)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for stop in ${STOPS:-1 3 104 4 0}; do
  export URMAPX_DEBUG_STOP=$stop
  rm -rf /tmp/pmc_$stop
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/pmc_$stop -o pmc --output-format csv -- python3 $R/bench.py --genome-mbp ${MBP:-800} --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pmc_$stop.log 2>&1
  echo "== stop $stop"; grep -o '"ms_per_step": [0-9.]*' /tmp/pmc_$stop.log | tail -1
  python3 $R/scripts/pmc_summary.py /tmp/pmc_$stop $R/gpurun_out/pmc_stop_$stop.json > /dev/null
done
