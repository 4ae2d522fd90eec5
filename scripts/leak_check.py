#!/usr/bin/env python3
"""GPU box: repeated urmapx_map_files calls (single-end and pairs, text phase and host phase) on one resident index; device
memory in use must come back to where it was after the first call of each kind."""
import gzip
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from urmap_amd import api

GOLD = os.path.join(ROOT, "tests", "golden")
d = tempfile.mkdtemp()
ufi = os.path.join(d, "g.ufi")
open(ufi, "wb").write(gzip.open(os.path.join(GOLD, "g.ufi.gz")).read())
torch.cuda.init()
idx = api.Index.open(ufi).upload(0)
# a bigger input: the golden reads 300 times over
fq = os.path.join(d, "big.fq")
with open(fq, "wb") as f:
    one = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    for _ in range(300):
        f.write(one)
f1, f2 = os.path.join(d, "b1.fq"), os.path.join(d, "b2.fq")
for src, dst in (("pe150_1.fq", f1), ("pe150_2.fq", f2)):
    with open(dst, "wb") as f:
        one = open(os.path.join(GOLD, src), "rb").read()
        for _ in range(300):
            f.write(one)


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 1e6


base = used()
log = []
for it in range(12):
    for kind in ("se_text", "pe_text", "se_host"):
        os.environ.pop("URMAPX_HOST_TEXT", None)
        if kind == "se_host":
            os.environ["URMAPX_HOST_TEXT"] = "1"
        if kind.startswith("se"):
            api.map_files(idx, fq, samout=os.path.join(d, "o.sam"), batch=20000, streams=2)
        else:
            api.map_files(idx, f1, f2, samout=os.path.join(d, "o2.sam"), batch=20000, streams=2)
    log.append(used())
print("device MB in use: before", round(base, 1), "after each round", [round(x, 1) for x in log])
assert max(log[2:]) - min(log[2:]) < 64, "device memory keeps growing"
api.lib().urmapx_host_pool_trim()
print("LEAK CHECK OK")
