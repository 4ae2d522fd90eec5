#!/bin/bash
# GPU box: kernel trace of the file-to-file legs (4 M reads: one file, null sink, two shards; two lanes, then one lane) -- where the
# GPU's time goes between the first and the last kernel of a urmapx_map_files run (scripts/e2e_timeline.py)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/r5t; mkdir -p $O
export URMAP_BENCH_NO_CLI=1 URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1 URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_E2E_READS=${E2E_READS:-4000000}
A="--steps 1 --warmup 1 --no-cpu-baseline --no-other-workloads"
python3 $R/bench.py $A > $O/plain.json 2> $O/plain.err   # builds the cache; the legs without the profiler
for lanes in 2 1; do
  URMAP_BENCH_E2E_STREAMS=$lanes timeout 900 rocprofv3 --kernel-trace -d /tmp/kt_e2e$lanes -o kt --output-format csv -- python3 $R/bench.py $A > $O/traced_lanes$lanes.json 2> $O/traced_lanes$lanes.err
  python3 $R/scripts/e2e_timeline.py /tmp/kt_e2e$lanes $O/timeline_lanes$lanes.txt
  rm -rf /tmp/kt_e2e$lanes
done
rm -rf /dev/shm/urmap_idx
python3 - <<PY
import json
for f in ("plain","traced_lanes2","traced_lanes1"):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); e=d["e2e"]
        print(f, "e2e", round(e["value"]/1e6,2), "null", round(e["null_sink"]["value"]/1e6,2), e["null_sink"]["stream_time_s_summed_over_lanes"], "sharded", round(e["sharded"]["value"]/1e6,2))
    except Exception as ex: print(f, ex)
PY
