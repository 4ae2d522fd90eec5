# GPU box: a light bench run whose last leg is the command line (index load included), with the loader's parts printed
mkdir -p gpurun_out/r5cli
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > gpurun_out/r5cli/light.json 2> gpurun_out/r5cli/light.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r5cli/light.json").read().strip().splitlines()[-1]); print(json.dumps(d["e2e"].get("cli"), indent=1))
PY
rm -rf /dev/shm/urmap_idx
