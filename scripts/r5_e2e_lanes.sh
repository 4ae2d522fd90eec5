#!/bin/bash
# GPU box: the file-to-file leg (10 M reads) with 2 / 3 / 4 lanes on the one GPU, twice (light bench runs on a cached index)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5l; mkdir -p $O
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1 URMAP_BENCH_NO_REFERENCE=1
A="--steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads"
for round in 1 2; do
for lanes in ${LANES:-2 3 4}; do
  URMAP_BENCH_E2E_STREAMS=$lanes python bench.py $A > $O/lanes${lanes}_$round.json 2> $O/lanes${lanes}_$round.err
  python - <<PY
import json
d=json.loads(open("$O/lanes${lanes}_$round.json").read().strip().splitlines()[-1]); e=d["e2e"]; n=e["null_sink"]
print("lanes=$lanes round $round: e2e", round(e["value"]/1e6,2), "null", round(n["value"]/1e6,2), n["seconds"], n["lanes_view"]["stream_time_s"], "busy", n["lane_busy_s_summed"], "sharded", round(e["sharded"]["value"]/1e6,2))
PY
done
done
rm -rf /dev/shm/urmap_idx
