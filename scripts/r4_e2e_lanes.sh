#!/bin/bash
# Round 4: the file-to-file legs (one file, text dropped, shards) with 2 / 3 / 4 mapping contexts (lanes) per GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
show() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
e=d['e2e']
print('$1', 'one file', e['value'], 'write ceiling', e['output_medium']['reads_per_s_at_that_ceiling'], 'null', e['null_sink']['value'], e['null_sink']['lanes'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', e['sharded']['value'], e['sharded']['vs_one_file'])"; }
for k in 1 2; do
for s in ${LANES:-2 3 4}; do
URMAP_BENCH_E2E_STREAMS=$s python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | show "lanes=$s"
done
done
rm -rf /dev/shm/urmap_idx
