# GPU box, round 6 call 18: two against three lanes once the process has 16 hardware queues (SAM text dropped, one file, two shards)
mkdir -p gpurun_out/r6q
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx GPU_MAX_HW_QUEUES=16
python scripts/r6_lanes.py 3100 10000000 files > gpurun_out/r6q/lanes_files_16q.txt 2>&1
grep -E "streams|one file|two shards" gpurun_out/r6q/lanes_files_16q.txt
rm -rf /dev/shm/urmap_idx
