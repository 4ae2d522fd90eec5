#!/bin/bash
# Round-2 GPU check: box facts, the whole -m gpu suite, then the hg38-scale bench (index cached in /dev/shm for later
# steps of the same call).  Outputs under gpurun_out/r2/.
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r2; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max; free -g | head -2; df -h /dev/shm /tmp | tail -2; rocm-smi --showmeminfo vram | grep -i total | head -2; python3 -c "import torch; print('devices', torch.cuda.device_count())"; } > $O/box.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
timeout 1200 python3 bench.py --steps 10 --warmup 2 > $O/bench_se.json 2> $O/bench_se.err; echo "bench rc=$?"
tail -c 1500 $O/bench_se.err; head -c 3000 $O/bench_se.json
rm -rf /dev/shm/urmap_idx
