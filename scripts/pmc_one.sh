#!/bin/bash
# diagnostic: SQ counters of one bench run (MBP, STOP optional) -> gpurun_out/pmc_one_$TAG.json
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
[ -n "$STOP" ] && export URMAPX_DEBUG_STOP=$STOP
rm -rf /tmp/pmc_one
timeout 600 rocprofv3 --pmc ${CTRS:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES} -d /tmp/pmc_one -o pmc --output-format csv -- python3 $R/bench.py --genome-mbp ${MBP:-800} --steps 2 --warmup 1 --no-cpu-baseline ${EXTRA_ARGS} > /tmp/pmc_one.log 2>&1
grep -o '"ms_per_step": [0-9.]*' /tmp/pmc_one.log | tail -1
python3 $R/scripts/pmc_summary.py /tmp/pmc_one $R/gpurun_out/pmc_one_${TAG:-x}.json > /dev/null
