# GPU box, round 6 call 4: the whole GPU suite on the library that ships (pair kernel on its LDS diet, chunk-size ramp in the text phase), the lanes with and
# without the ramp, and the pair workload under rocprofv3 --kernel-trace --stats on the round-5 pair kernel and on the diet (what the second pass costs either)
mkdir -p gpurun_out/r6d
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6d/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r6d/pytest_gpu.txt
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
python scripts/r6_lanes.py 3100 10000000 ramp > gpurun_out/r6d/lanes_ramp.txt 2>&1
grep -E "streams|one file|two shards" gpurun_out/r6d/lanes_ramp.txt
R=$PWD; cd /tmp; export TMPDIR=/tmp
A="--steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e --mode pe"
for v in diet base; do
  if [ $v = diet ]; then unset URMAPX_LIB; else export URMAPX_LIB=$R/urmap_amd/csrc/build_nopf/liburmapx.so; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_pe_$v -o kt --output-format csv -- python3 $R/bench.py $A > $R/gpurun_out/r6d/bench_pe_${v}_ktrace.json 2> $R/gpurun_out/r6d/kt_pe_$v.err
  cp $(find /tmp/kt_pe_$v -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6d/kernel_stats_pe_$v.csv
  echo "== $v"; head -6 $R/gpurun_out/r6d/kernel_stats_pe_$v.csv | cut -c1-150
done
unset URMAPX_LIB
cd $R
rm -rf /dev/shm/urmap_idx
