# GPU box, round 6 call 13: what slows bench.py's null-sink leg against scripts/r6_lanes.py on the same box?  (idle contexts on the device, the oracle's OpenMP team)
mkdir -p gpurun_out/r6m
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python scripts/r6_lanes.py 3100 10000000 idle > gpurun_out/r6m/lanes_idle.txt 2>&1
grep -E "M reads/s" gpurun_out/r6m/lanes_idle.txt
rm -rf /dev/shm/urmap_idx
