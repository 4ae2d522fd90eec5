#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in a hipcc -save-temps .s file (the .amdhsa metadata at its end)."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = re.findall(r'\.group_segment_fixed_size: (\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size: (\d+).*?\.sgpr_count:\s+(\d+)'
                  r'.*?\.vgpr_count:\s+(\d+)', s, re.S)
names = subprocess.run(['c++filt'], input="\n".join(r[1] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (lds, _, scr, sg, vg), name in zip(rows, names):
    name = name.split('(')[0]
    if pat in name:
        print(f"{name:60s} lds {lds:>6s} scratch {scr:>5s} sgpr {sg:>4s} vgpr {vg:>4s}")
