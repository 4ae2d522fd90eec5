R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2
rocprofv3 -L > $R/gpurun_out/r2/counters_list.txt 2>&1
grep -i -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*" $R/gpurun_out/r2/counters_list.txt | sort -u | head -40
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_IFETCH -d /tmp/pi -o pi --output-format csv -- python3 $R/scripts/stop_sweep.py 800 150 0.01 0.001 1000000 3 0 > $R/gpurun_out/r2/icache_run.txt 2>&1
tail -5 $R/gpurun_out/r2/icache_run.txt
python3 $R/scripts/pmc_summary.py /tmp/pi $R/gpurun_out/r2/pmc_icache.json | head -80
