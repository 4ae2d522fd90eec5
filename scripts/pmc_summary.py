#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run: per kernel and counter, the average counter value per dispatch.

usage: pmc_summary.py <rocprofv3 output dir> [out.json]
Reads every *counter_collection.csv under the directory (columns Kernel_Name, Counter_Name, Counter_Value).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].split("(")[0]
                full = row["Kernel_Name"]
                for short in ("search_se_kernel", "search_pe_kernel", "seed_probe_kernel", "viterbi_batch_kernel", "dp_kernel", "finalize_se_kernel"):
                    if short in k:
                        # <NCH, true> = the second pass over reads whose lists outgrew LDS (usually an empty queue)
                        k = short + ("_pass2" if re.search(r"<\d+,\s*true", full) else "") + ("_dbg" if re.search(r"<\d+,\s*(true|false),\s*true", full) else "")
                        m = re.search(r"<\d+,\s*(?:true|false),\s*(?:true|false),\s*(?:true|false),\s*([12])>", full)  # round 5: phase 3 parked -- first / second launch
                        if m:
                            k += "_part" + m.group(1)
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    out = {k: {c: {"avg": v[0] / v[1], "dispatches": v[1], "sum": v[0]} for c, v in cs.items()} for k, cs in acc.items()
           if k.replace("_pass2", "").replace("_dbg", "").replace("_part1", "").replace("_part2", "") in ("search_se_kernel", "search_pe_kernel", "seed_probe_kernel", "viterbi_batch_kernel", "dp_kernel", "finalize_se_kernel")}
    s = json.dumps(out, indent=1, sort_keys=True)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(s + "\n")
    print(s)


if __name__ == "__main__":
    main()
