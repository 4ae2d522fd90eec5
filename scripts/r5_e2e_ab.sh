#!/bin/bash
# GPU box: the file-to-file legs incl. gz / BGZF / pairs with the round-5 lane changes switched off one by one (light bench runs, one box)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5ab; mkdir -p $O
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1
A="--steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads"
run() {  # name, env...
  name=$1; shift
  env "$@" python bench.py $A > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); e=d["e2e"]
g=e["gz"]; p=e["pairs"]
print("$name: e2e", round(e["value"]/1e6,2), "null", round(e["null_sink"]["value"]/1e6,2), "sharded", round(e["sharded"]["value"]/1e6,2),
      "gzip", round(g["gzip"]["value"]/1e6,2), g["gzip"]["inflate_GBs"], g["gzip"]["lanes_view"]["alloc"], "bgzf", round(g["bgzf"]["value"]/1e6,2), g["bgzf"]["inflate_GBs"], g["bgzf"]["lanes_view"]["alloc"],
      "pairs", round(p["value"]/1e6,2), p["stage_busy_s"], p["lanes_view"]["alloc"])
PY
}
for round in 1 2; do
run default_$round A=1
run no_deferred_$round URMAPX_NO_DEFERRED_COPY=1
run no_pool_$round URMAPX_NO_LANE_POOL=1
run neither_$round URMAPX_NO_DEFERRED_COPY=1 URMAPX_NO_LANE_POOL=1
done
rm -rf /dev/shm/urmap_idx
