R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
O=$R/gpurun_out/full; mkdir -p $O
timeout 900 python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 5 --warmup 1 > $O/bench_se250.json 2> $O/bench_se250.err
timeout 900 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf250 -o pf --output-format csv -- python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se250_pmcfetch.json 2> $O/pf250.err
python3 $R/scripts/pmc_summary.py /tmp/pf250 $O/pmc_fetch_se250_raw.json > /dev/null
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt250 -o kt --output-format csv -- python3 $R/bench.py --read-len 250 --sub 0.04 --indel 0.01 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_se250_kt.json 2> $O/kt250.err
cp $(find /tmp/kt250 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_se250.csv
rm -rf /dev/shm/urmap_idx
tail -c 400 $O/bench_se250.json; cat $O/pmc_fetch_se250_raw.json
