# GPU box, round 6 call 14: bench.py's order of file-to-file calls replayed in scripts/r6_lanes.py: what makes the null-sink calls slower there?
mkdir -p gpurun_out/r6n
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python scripts/r6_lanes.py 3100 10000000 idle files_first > gpurun_out/r6n/lanes_order.txt 2>&1
grep -E "M reads/s" gpurun_out/r6n/lanes_order.txt
rm -rf /dev/shm/urmap_idx
