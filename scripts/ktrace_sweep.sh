R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/scripts/stop_sweep.py ${MBP:-800} ${L:-150} ${SUB:-0.01} ${INDEL:-0.001} 1000000 0 > $R/gpurun_out/r2/kt_run.txt 2>&1
grep "production\|stop" $R/gpurun_out/r2/kt_run.txt
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r2/kt_stats_${MBP:-800}_${L:-150}.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "search_se" in n or "seed_probe" in n:
        print(n.split("(")[0][:60], r["Calls"], "avg_ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
PY
