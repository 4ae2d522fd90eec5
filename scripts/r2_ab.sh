#!/bin/bash
# A/B of a build-time knob: scripts/r2_ab.sh "<make EXTRA flags A>" "<flags B>" ...   (headline + 250 bp, kernels only)
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for flags in "$@"; do
  (cd urmap_amd/csrc && touch kernels.hip kernels.h && make -j16 EXTRA="$flags" > /dev/null 2>&1)
  python3 bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$flags', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:6],k['avg_ms']) for k in d['kernels']])
for n,v in d['other_workloads'].items(): print('   ', n, v['value'], v['parity']['bit_identical_to_oracle'], [(k['kernel'][:6],k['avg_ms']) for k in v['kernels']])
"
done
rm -rf /dev/shm/urmap_idx
