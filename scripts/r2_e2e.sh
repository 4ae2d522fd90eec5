#!/bin/bash
# default bench without the other workloads: headline + e2e (single-end and pairs), with the pipeline trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r2
URMAPX_PIPE_TRACE=1 timeout 2000 python3 bench.py --no-other-workloads > gpurun_out/r2/bench_e2e_pairs.json 2> gpurun_out/r2/bench_e2e_pairs.err
echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/bench_e2e_pairs.json").read().strip().splitlines()[-1])
e=d["e2e"]; print(d["value"]); print({k:v for k,v in e.items() if k not in ("reference_binary","pairs")}); print(e.get("pairs"))
PY
grep "^trace" gpurun_out/r2/bench_e2e_pairs.err | tail -56
grep -v "^trace" gpurun_out/r2/bench_e2e_pairs.err | tail -5
