#!/bin/bash
# Round 4: A/B of library builds on one box (index cached in /dev/shm across the runs).
# usage: r4_ab.sh lib1.so lib2.so ...   env: WL = se150 | pe | se250 (default se150), ROUNDS, EXTRA (bench.py flags), TESTS (pytest -k expression, run first on the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
if [ -n "$TESTS" ]; then
  timeout 1200 python3 -m pytest tests -m gpu -x -q -k "$TESTS" > gpurun_out/r4/pytest_ab.txt 2>&1
  tail -5 gpurun_out/r4/pytest_ab.txt | cut -c1-300
fi
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for wl in ${WL:-se150}; do
case $wl in
  se150) W="";;
  pe) W="--mode pe";;
  se250) W="--read-len 250 --sub 0.04 --indel 0.01";;
esac
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  URMAPX_LIB=$PWD/$lib timeout 900 python3 bench.py $W --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $EXTRA 2>gpurun_out/r4/ab_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', '$lib', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']])"
done
done
done
rm -rf /dev/shm/urmap_idx
