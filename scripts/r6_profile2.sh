#!/bin/bash
# Round-6 profiles of the code that ships at the end of the round (code version r6b: search_se_kernel<3, .., KCH 2>, row stores in LDS, PosToCoordL by lanes), hg38 scale,
# index cached in /dev/shm across the passes of this one call.  Per workload (single-end 150 = headline, pairs 2x150, single-end 250 / 5 %): rocprofv3 --kernel-trace --stats
# with one context (the launch with the device to itself: what roofline.achieved is priced on) and, for the headline, with the default two; --pmc FETCH_SIZE, --pmc WRITE_SIZE
# and two --pmc SQ_* passes, counters always in runs of their own, one context.
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export URMAP_BENCH_INDEX_CACHE=/dev/shm/urmap_idx
V=${V:-r6b}
O=$R/gpurun_out/$V/prof; mkdir -p $O
T="timeout 900"
for wl in ${WORKLOADS:-se150 pe se250}; do
  case $wl in
    se150) W="";;
    pe) W="--mode pe";;
    se250) W="--read-len 250 --sub 0.04 --indel 0.01";;
  esac
  A="--steps 5 --warmup 1 --contexts 1 --no-cpu-baseline --no-other-workloads --no-e2e $W"
  $T rocprofv3 --kernel-trace --stats -d /tmp/kt_$wl -o kt --output-format csv -- python3 $R/bench.py $A > $O/bench_${wl}_ktrace.json 2> $O/kt_$wl.err
  cp $(find /tmp/kt_$wl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${wl}_$V.csv
  if [ $wl = se150 ]; then
    A2="--steps 5 --warmup 1 --no-cpu-baseline --no-other-workloads --no-e2e $W"
    $T rocprofv3 --kernel-trace --stats -d /tmp/kt2_$wl -o kt --output-format csv -- python3 $R/bench.py $A2 > $O/bench_${wl}_ktrace_two_contexts.json 2> $O/kt2_$wl.err
    cp $(find /tmp/kt2_$wl -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${wl}_two_contexts_$V.csv
    rm -rf /tmp/kt2_$wl
  fi
  $T rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$wl -o pf --output-format csv -- python3 $R/bench.py $A > $O/bench_${wl}_pmcfetch.json 2> $O/pf_$wl.err
  python3 $R/scripts/pmc_summary.py /tmp/pf_$wl $O/pmc_fetch_${wl}_raw.json > /dev/null
  $T rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$wl -o pw --output-format csv -- python3 $R/bench.py $A > $O/bench_${wl}_pmcwrite.json 2> $O/pw_$wl.err
  python3 $R/scripts/pmc_summary.py /tmp/pw_$wl $O/pmc_write_${wl}_raw.json > /dev/null
  $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES -d /tmp/ps_$wl -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_${wl}_pmcsq.json 2> $O/ps_$wl.err
  python3 $R/scripts/pmc_summary.py /tmp/ps_$wl $O/pmc_sq_${wl}.json > /dev/null
  $T rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT -d /tmp/ps2_$wl -o ps --output-format csv -- python3 $R/bench.py $A > $O/bench_${wl}_pmcsq2.json 2> $O/ps2_$wl.err
  python3 $R/scripts/pmc_summary.py /tmp/ps2_$wl $O/pmc_sq2_${wl}.json > /dev/null
  case $wl in
    se150) M="se 150";;
    pe) M="pe 150";;
    se250) M="se 250";;
  esac
  python3 $R/scripts/pmc_fetch_json.py $O/pmc_fetch_${wl}_raw.json $O/pmc_fetch_hg38scale_${wl}_$V.json $M 3100000727 $V 6 "--contexts 1 $W" FETCH_SIZE > /dev/null
  python3 $R/scripts/pmc_fetch_json.py $O/pmc_write_${wl}_raw.json $O/pmc_write_hg38scale_${wl}_$V.json $M 3100000727 $V 6 "--contexts 1 $W" WRITE_SIZE > /dev/null
  rm -rf /tmp/kt_$wl /tmp/pf_$wl /tmp/pw_$wl /tmp/ps_$wl /tmp/ps2_$wl
  echo "== $wl"; head -8 $O/kernel_stats_${wl}_$V.csv | cut -c1-150
  python3 - <<PY
import json
for f in ("pmc_fetch_${wl}_raw","pmc_write_${wl}_raw","pmc_sq_$wl","pmc_sq2_$wl"):
    try: d=json.load(open("$O/"+f+".json"))
    except Exception as e: print(f, e); continue
    for k,v in d.items(): print(f, k, {c: round(x['avg']/1e6,1) for c,x in v.items()})
PY
done
rm -rf /dev/shm/urmap_idx
