#!/usr/bin/env python3
"""GPU timeline of the file-to-file legs from a rocprofv3 --kernel-trace run of bench.py (kernel_trace.csv: one row per
dispatch with its start and end): for every urmapx_map_files run (a burst of text kernels), the wall from its first to its
last kernel, the time at least one kernel was running (union), the idle gaps, and the summed duration per kernel name.

usage: e2e_timeline.py <rocprofv3 output dir> [out.txt]
"""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    n = name
    for p in ("void ", "urx::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("(")[0].strip()


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "")))
    rows.sort()
    text = [r for r in rows if r[2].startswith(("nl_count", "sam_kernel", "sam_len"))]
    if not text:
        print("no text kernels in the trace")
        return
    # runs: bursts of text kernels separated by more than 100 ms
    runs, cur = [], [text[0]]
    for r in text[1:]:
        if r[0] - cur[-1][1] > 100_000_000:
            runs.append(cur); cur = [r]
        else:
            cur.append(r)
    runs.append(cur)
    out = []
    for k, run in enumerate(runs):
        t0, t1 = run[0][0], max(r[1] for r in run)
        inside = [r for r in rows if r[0] >= t0 and r[1] <= t1]
        busy, end, gaps = 0, t0, []
        for s, e, _, _ in inside:
            if s > end:
                gaps.append(s - end); busy += e - s; end = e
            elif e > end:
                busy += e - end; end = e
        per = defaultdict(lambda: [0, 0])
        for s, e, n, _ in inside:
            per[n][0] += e - s; per[n][1] += 1
        n_sam = sum(1 for r in inside if r[2].startswith("sam_kernel"))
        out.append(f"run {k}: wall {(t1 - t0) / 1e6:.1f} ms, some kernel running {busy / 1e6:.1f} ms ({100.0 * busy / (t1 - t0):.0f} %), "
                   f"{len(gaps)} gaps, {sum(gaps) / 1e6:.1f} ms idle (largest {max(gaps or [0]) / 1e6:.2f} ms), {n_sam} chunks, queues {len(set(r[3] for r in inside))}")
        for n, (d, c) in sorted(per.items(), key=lambda x: -x[1][0])[:14]:
            out.append(f"    {n[:70]:70s} {c:5d} launches {d / 1e6:9.2f} ms summed {d / c / 1e3:9.1f} us each")
    s = "\n".join(out)
    print(s)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")


if __name__ == "__main__":
    main()
