#!/bin/bash
# the single-end step with 16 / 12 / 8 / 4 resident blocks (= waves) per CU of the same search kernel binary
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for b in ${BLOCKS:-16 14 12 10 8 4}; do
  URMAPX_TEST_BLOCKS_PER_CU=$b python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $EXTRA 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('blocks/CU $b', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']][:2])"
done
rm -rf /dev/shm/urmap_idx
