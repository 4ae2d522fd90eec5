#!/bin/bash
# Round 4: the chain-row layout on / off with ONE library on one box (URMAPX_NO_CHAIN_ROWS=1: the kernels walk the chains hop by hop)
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for wl in ${WL:-se150 se250}; do
case $wl in
  se150) W="";;
  pe) W="--mode pe";;
  se250) W="--read-len 250 --sub 0.04 --indel 0.01";;
esac
for round in 1 2; do
for rows in 0 1; do
  if [ $rows = 0 ]; then export URMAPX_NO_CHAIN_ROWS=1; else unset URMAPX_NO_CHAIN_ROWS; fi
  timeout 900 python3 bench.py $W --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'rows=$rows', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']][:3], d['config']['setup_s'])"
done
done
done
rm -rf /dev/shm/urmap_idx
