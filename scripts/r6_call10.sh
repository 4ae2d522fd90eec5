# GPU box, round 6 call 10: the single-end step against the batch size, and whole batches alternating over two contexts (scripts/r6_pingpong.py)
mkdir -p gpurun_out/r6j
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python scripts/r6_pingpong.py 3100 > gpurun_out/r6j/pingpong.txt 2>&1
grep -E "context" gpurun_out/r6j/pingpong.txt
rm -rf /dev/shm/urmap_idx
