#!/bin/bash
# round 2: text path timeline (URMAPX_PIPE_TRACE) and its kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/text2
export TMPDIR=/tmp
URMAPX_PIPE_TRACE=1 timeout 900 python scripts/e2e_probe.py --genome-mbp 400 --reads 4000000 --repeat 2 --set text:2:262144 \
  > gpurun_out/text2/probe_trace.log 2> gpurun_out/text2/probe_trace.err
grep -v "^trace" gpurun_out/text2/probe_trace.err | tail -3
cat gpurun_out/text2/probe_trace.log
grep "^trace" gpurun_out/text2/probe_trace.err | tail -70
cd /tmp
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/text2/ktrace" -o text -- python3 "$GRAFT_REPO_ROOT/scripts/e2e_probe.py" --genome-mbp 400 --reads 4000000 --repeat 1 --set text:2:262144 > "$GRAFT_REPO_ROOT/gpurun_out/text2/ktrace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/text2/ktrace -name "*stats*" | head
f=$(find gpurun_out/text2/ktrace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" | cut -c1-200
f=$(find gpurun_out/text2/ktrace -name "*memory_copy_stats.csv" | head -1)
[ -n "$f" ] && cat "$f" | cut -c1-200
# keep the big traces out of the merge
find gpurun_out/text2/ktrace -name "*trace.csv" -size +20M -delete
