#!/usr/bin/env python3
"""GPU box: random-access slot rate of the resident table by access shape (URMAPX_GATHER_MODE, kernels.hip gather_bench_kernel)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from urmap_amd import api, ranks
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
slots_n = int(5392814809 * mbp / 3100) | 1
blob = torch.zeros(5 * slots_n + 64, dtype=torch.uint8, device=dev)
seq = torch.zeros(4096 + 1000, dtype=torch.uint8, device=dev)
import numpy as np
idx = api.Index.wrap_device(0, 24, 32, slots_n, blob.data_ptr(), seq.data_ptr(), 1000, np.array([1000], np.uint32), np.array([0], np.uint32), ["x"])
m = api.Mapper(idx, device=0)
names = {0: "one 8-byte load per slot", 1: "two 4-byte loads per slot", 2: "two 4-byte LDS-DMA loads per slot", 3: "one 12-byte LDS-DMA load per slot", 4: "one aligned 16-byte load per slot"}
for mode in (0, 1, 2, 3, 4, 0):
    os.environ["URMAPX_GATHER_MODE"] = str(mode)
    r = m.gather_microbench(1 << 28)
    print(f"mode {mode} ({names[mode]}): {r / 1e9:.2f} G slots/s = {64 * r / 1e12:.2f} TB/s of sectors")
