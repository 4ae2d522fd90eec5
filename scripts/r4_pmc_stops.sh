#!/bin/bash
# Per-stage instruction counts of the search_se_kernel (final round-4 code) at hg38 scale: the diagnostic instantiation with the
# schedule cut after step N (URMAPX_DEBUG_STOP), one rocprofv3 --pmc run per cut; differences between consecutive cuts
# = the stage's share.  usage: r3_pmc_stops.sh [lib.so]
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
[ -n "$1" ] && export URMAPX_LIB=$R/$1
O=$R/gpurun_out/r4/stops_${TAG:-v1}; mkdir -p $O
python3 $R/scripts/stop_sweep.py 3100 150 0.01 0.001 1000000 100 1 3 104 4 0 2>&1 | grep -v amdgpu.ids > $O/sweep_ms.txt
grep "production\|stop" $O/sweep_ms.txt
for stop in 100 1 3 104 4 0; do
  rm -rf /tmp/pmc_$stop
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA -d /tmp/pmc_$stop -o pmc --output-format csv -- python3 $R/scripts/stop_sweep.py 3100 150 0.01 0.001 1000000 $stop > /tmp/pmc_$stop.log 2>&1
  python3 $R/scripts/pmc_summary.py /tmp/pmc_$stop $O/pmc_stop_$stop.json > /dev/null
done
rm -rf /dev/shm/urmap_idx
python3 - <<PY
import json
for stop in (100,1,3,104,4,0):
    d=json.load(open(f'$O/pmc_stop_{stop}.json')).get('search_se_kernel_dbg',{})
    print(stop, {k: round(v['avg']/1e6,1) for k,v in d.items()})
PY
