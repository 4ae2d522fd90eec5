#!/bin/bash
# per-dispatch durations of the phase-6 launches (dp_kernel / finalize_se_kernel), 150 bp and 250 bp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/dptrace
export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "150 0.01 0.001" "250 0.04 0.01"; do
  set -- $cfg
  rocprofv3 --kernel-trace -d /tmp/dpt_$1 -o t --output-format csv -- python3 $R/bench.py --no-e2e --no-other-workloads --no-cpu-baseline --steps 3 --warmup 1 --read-len $1 --sub $2 --indel $3 > $R/gpurun_out/dptrace/bench_$1.json 2> $R/gpurun_out/dptrace/err_$1.txt
  f=$(find /tmp/dpt_$1 -name "*kernel_trace.csv" | head -1)
  python3 - "$f" $1 <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
sel=[r for r in rows if 'dp_kernel' in r['Kernel_Name'] or 'finalize_se' in r['Kernel_Name'] or 'search_se' in r['Kernel_Name'] or 'seed_probe' in r['Kernel_Name']]
sel.sort(key=lambda r:int(r['Start_Timestamp']))
# last step only: the last 20 launches
out=[]
for r in sel[-22:]:
    n=r['Kernel_Name']
    short='dp' if 'dp_kernel' in n else 'fin' if 'finalize' in n else 'search' if 'search_se' in n else 'probe'
    out.append((short,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, r.get('Grid_Size','')))
print("L=%s:"%sys.argv[2], " ".join("%s %.2f"%(a,b) for a,b,_ in out))
PY
done
rm -rf /dev/shm/urmap_idx
