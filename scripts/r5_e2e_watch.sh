#!/bin/bash
# GPU box: full default bench runs with the host side of every file-to-file leg watched (cgroup throttling, process CPU seconds, the lanes'
# map time split into search / dp / host enqueue), then the kernel timeline of the legs (scripts/r5_e2e_trace.sh)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5w; mkdir -p $O
cat /sys/fs/cgroup/cpu.max > $O/box.txt 2>&1; nproc >> $O/box.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for k in ${RUNS:-1 2 3}; do
  URMAPX_PIPE_TRACE=1 python bench.py > $O/full$k.json 2> $O/full$k.err
  python - <<PY
import json
d=json.loads(open("$O/full$k.json").read().strip().splitlines()[-1]); e=d["e2e"]
print("run $k:", d["value"], d["ms_per_step"], "e2e", round(e["value"]/1e6,2), "first", e["first_run_seconds"], "null", round(e["null_sink"]["value"]/1e6,2), "sharded", round(e["sharded"]["value"]/1e6,2), "gz", round(e["gz"]["gzip"]["value"]/1e6,2), "pairs", round(e["pairs"]["value"]/1e6,2))
print("   first", e["first_run_lanes_view"]); print("   e2e  ", e["lanes_view"]); print("   null ", e["null_sink"]["lanes_view"]); print("   shard", e["sharded"]["lanes_view"])
PY
done
if [ -z "$NO_TRACE" ]; then bash scripts/r5_e2e_trace.sh; fi
