#!/bin/bash
# Round 4: PE tests on the current library, then A/B of library builds on the pair workload on one box.
# usage: r4_pe_ab.sh lib1.so lib2.so ...   env: SKIP_TESTS, ROUNDS, TESTS (pytest -k expression)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
if [ -z "$SKIP_TESTS" ]; then
  timeout 900 python3 -m pytest tests -m gpu -x -q -k "${TESTS:-pe or map2 or pair or rescue}" > gpurun_out/r4/pytest_pe.txt 2>&1
  tail -5 gpurun_out/r4/pytest_pe.txt | cut -c1-300
fi
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  URMAPX_LIB=$PWD/$lib timeout 600 python3 bench.py --mode pe --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $EXTRA 2>gpurun_out/r4/ab_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']])"
done
done
rm -rf /dev/shm/urmap_idx
