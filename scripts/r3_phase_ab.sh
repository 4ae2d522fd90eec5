#!/bin/bash
# per-phase cycles of the search kernel for several builds of the library on one box: r3_phase_ab.sh lib1.so lib2.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
for lib in "$@"; do
  URMAPX_LIB=$lib python3 scripts/phase_ab.py 3100 1000000 2>&1 | grep -v amdgpu.ids
done
for lib in "$@"; do
  URMAPX_LIB=$lib python3 bench.py --no-cpu-baseline --no-other-workloads --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['value'], d['ms_per_step'], d['parity']['bit_identical_to_oracle'], [(k['kernel'][:9],k['avg_ms']) for k in d['kernels']])"
done
rm -rf /dev/shm/urmap_idx
