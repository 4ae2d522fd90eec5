#!/bin/bash
# PC sampling of the single-end search kernel (rocprofv3 beta feature; host-trap method): where the waves' program
# counters are, by code-object offset.  Small genome by default: the hot spots do not depend on the table size.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r3/pcs; mkdir -p $O
A="--genome-mbp ${MBP:-400} --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $EXTRA"
timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval ${INTERVAL:-200} \
  -d /tmp/pcs -o pcs --output-format csv -- python3 $R/bench.py $A > $O/bench.json 2> $O/err.txt
echo "rc=$?"; tail -3 $O/err.txt | cut -c1-300
ls -la /tmp/pcs 2>/dev/null | head; find /tmp/pcs -type f | head
for f in $(find /tmp/pcs -name "*pc_sampling*csv" | head -2); do head -5 $f; wc -l $f; done
python3 - <<'PY'
import csv, glob, collections, os
fs = glob.glob('/tmp/pcs/**/*pc_sampling_host_trap.csv', recursive=True) + glob.glob('/tmp/pcs/**/*pc_sampling*.csv', recursive=True)
if fs:
    f = fs[0]
    c = collections.Counter()
    n = 0
    with open(f) as fh:
        rd = csv.DictReader(fh)
        cols = rd.fieldnames
        print(cols)
        for r in rd:
            n += 1
            c[(r.get('Code_Object_Id'), r.get('Code_Object_Offset') or r.get('Instruction'))] += 1
    out = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out/r3/pcs/hist.txt'
    with open(out, 'w') as o:
        o.write(f"# {n} samples from {f}\n")
        for (co, off), k in c.most_common(4000):
            o.write(f"{co}\t{off}\t{k}\n")
    print(n, 'samples;', len(c), 'distinct pcs')
PY
