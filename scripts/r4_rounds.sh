#!/bin/bash
# Round 4: phase 6's launches one by one (bench.py phase6.launch_ms_by_round) for the 150- and the 250-base workload
# env: WL (se150 se250), LIBS (library builds to compare, default the in-tree one), TESTS=1 (GPU suite first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
if [ -n "$TESTS" ]; then
  timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4/pytest_rounds.txt 2>&1
  tail -5 gpurun_out/r4/pytest_rounds.txt | cut -c1-300
fi
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for wl in ${WL:-se150 se250}; do
case $wl in
  se150) W="";;
  se250) W="--read-len 250 --sub 0.04 --indel 0.01";;
esac
for lib in ${LIBS:-urmap_amd/liburmapx.so}; do
URMAPX_LIB=$PWD/$lib timeout 900 python3 bench.py $W --steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e 2>gpurun_out/r4/rounds_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', '$lib', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']][:3], json.dumps(d['phase6']['launch_ms_by_round']))"
done
done
rm -rf /dev/shm/urmap_idx
