mkdir -p gpurun_out/r5a
{ free -g; df -h /dev/shm /tmp; nproc; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/cpu.max 2>/dev/null; } > gpurun_out/r5a/box.txt 2>&1
python -m pytest tests/test_gpu_validate.py -x -q -m gpu > gpurun_out/r5a/validate_tests.txt 2>&1
python scripts/r5_ufi_fullscale.py --mbp 100 --out gpurun_out/r5a/ufi_100mbp --ref-validate > gpurun_out/r5a/ufi_100mbp.log 2>&1
( time python -m pytest tests/test_gpu_fullscale.py -x -q -m gpu ) > gpurun_out/r5a/fullscale_tests.txt 2>&1
python bench.py > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err
tail -3 gpurun_out/r5a/validate_tests.txt gpurun_out/r5a/fullscale_tests.txt; tail -c 600 gpurun_out/r5a/ufi_100mbp.log
