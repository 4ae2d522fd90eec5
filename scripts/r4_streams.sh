#!/bin/bash
# Round 4: does a search grid that leaves room on the CUs let the other context's DP / finalize launches run beside it?
# bench.py --streams 2 with 16 .. 12 resident search blocks per CU (URMAPX_TEST_BLOCKS_PER_CU), and --streams 1 for reference.
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
run() {
  python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']][:3])"
}
run "--streams 1" "streams 1, 16 blocks/CU"
for b in 16 15 14 13 12 10; do
  URMAPX_TEST_BLOCKS_PER_CU=$b run "--streams 2" "streams 2, $b blocks/CU"
done
for b in 14 12; do
  URMAPX_TEST_BLOCKS_PER_CU=$b run "--streams 3" "streams 3, $b blocks/CU"
done
rm -rf /dev/shm/urmap_idx
