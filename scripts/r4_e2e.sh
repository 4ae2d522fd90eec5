#!/bin/bash
# Round 4: the file-to-file legs only (one file, text dropped, shards), one rank and two ranks on the box's GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx URMAP_BENCH_NO_REFERENCE=1 URMAP_BENCH_NO_E2E_GZ=1 URMAP_BENCH_NO_E2E_PAIRS=1
show() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
e=d['e2e']
print('$1', 'one file', e['value'], 'write ceiling', e['output_medium']['reads_per_s_at_that_ceiling'], 'null', e['null_sink']['value'], e['null_sink']['stream_time_s_summed_over_lanes'], 'sharded', e['sharded']['value'], e['sharded']['vs_one_file'], e['sharded']['cat_of_shards_equals_the_one_file'])"; }
for k in 1 2; do
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | show "1 rank"
URMAP_BENCH_E2E_READS=4000000 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads 2>/dev/null | show "2 ranks"
done
rm -rf /dev/shm/urmap_idx
