#!/bin/bash
# GPU box: the tests that go through the command line and the loaders, then a light bench run whose last leg is the command line itself (index load included)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5cli; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_text.py tests/test_gpu_validate.py tests/test_gpu_slow.py -x -q -m gpu 2>&1 | tail -6
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/light.json 2> $O/light.err
tail -3 $O/light.err
python - <<PY
import json
d=json.loads(open("$O/light.json").read().strip().splitlines()[-1]); e=d["e2e"]
print("e2e", round(e["value"]/1e6,2), "null", round(e["null_sink"]["value"]/1e6,2))
print(json.dumps(e.get("cli"), indent=1))
PY
rm -rf /dev/shm/urmap_idx
