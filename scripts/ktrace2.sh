# kernel trace of the production kernels on two workloads (no debug stops): first/second pass durations
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2
for cfg in "150 0.01 0.001" "250 0.04 0.01"; do
  set -- $cfg
  rm -rf /tmp/kt
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- python3 $R/scripts/stop_sweep.py ${MBP:-800} $1 $2 $3 1000000 0 > $R/gpurun_out/r2/kt_run_$1.txt 2>&1
  grep "production\|stop" $R/gpurun_out/r2/kt_run_$1.txt
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r2/kt_stats_${MBP:-800}_$1.csv
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "search_se" in n or "seed_probe" in n:
        print(n.split("(")[0][:60], r["Calls"], "avg_ms %.2f" % (float(r["AverageNs"])/1e6), "max %.2f" % (float(r["MaxNs"])/1e6))
PY
done
