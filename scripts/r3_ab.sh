#!/bin/bash
# A/B of library builds on one box (index cached in /dev/shm): r3_ab.sh lib1.so lib2.so ...   env: EXTRA = bench.py flags, ROUNDS
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  URMAPX_LIB=$lib python3 bench.py --no-cpu-baseline --no-other-workloads --no-e2e $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']])"
done
done
rm -rf /dev/shm/urmap_idx
