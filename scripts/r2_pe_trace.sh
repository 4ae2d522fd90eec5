R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/r2
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ktp -o kt --output-format csv -- python3 $R/bench.py --mode pe --genome-mbp ${MBP:-3100} --steps 4 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e > $R/gpurun_out/r2/bench_pe_ktrace.json 2> /tmp/ktp.err
f=$(find /tmp/ktp -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r2/kernel_stats_pe.csv
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "search_pe" in n or "seed_probe" in n:
        print(n.split("(")[0][:60], r["Calls"], "avg_ms %.2f" % (float(r["AverageNs"])/1e6), "max %.2f" % (float(r["MaxNs"])/1e6))
PY
python3 -c "
import json
d=json.loads(open('$R/gpurun_out/r2/bench_pe_ktrace.json').read().strip().splitlines()[-1]); print(d['value'], d['parity'], d['work_per_read'])"
