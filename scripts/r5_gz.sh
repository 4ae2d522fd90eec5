#!/bin/bash
# GPU box: the .gz reader's phases on the host (scripts/pgzip_phases.py), then the gz / BGZF legs of bench.py twice (light runs)
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r5gz; mkdir -p $O
[ -z "$NO_PHASES" ] && python scripts/pgzip_phases.py 2000000 16 2>&1 | tail -12
[ -z "$NO_PHASES" ] && URMAPX_PGZIP_NO_SIMD=1 python scripts/pgzip_phases.py 2000000 16 2>&1 | tail -6
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_PAIRS=1
for k in 1 2 3; do
  for simd in on off; do
    if [ $simd = off ]; then export ${OFFVAR:-URMAPX_PGZIP_NO_SIMD}=1; else unset ${OFFVAR:-URMAPX_PGZIP_NO_SIMD}; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads > $O/gz_${simd}_$k.json 2> $O/gz_${simd}_$k.err
    python - <<PY
import json
d=json.loads(open("$O/gz_${simd}_$k.json").read().strip().splitlines()[-1]); g=d["e2e"]["gz"]
print("simd $simd run $k: gzip", round(g["gzip"]["value"]/1e6,2), g["gzip"]["inflate_GBs"], "bgzf", round(g["bgzf"]["value"]/1e6,2), g["bgzf"]["inflate_GBs"], "e2e", round(d["e2e"]["value"]/1e6,2))
PY
  done
done
rm -rf /dev/shm/urmap_idx
