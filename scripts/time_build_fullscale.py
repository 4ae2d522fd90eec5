import sys, time, os
sys.path.insert(0, '.')
import numpy as np, torch
from urmap_amd import api
G = int(float(sys.argv[1]) * 1e6)
t = time.time()
g = torch.Generator(device='cuda'); g.manual_seed(1)
d = torch.randint(0, 4, (G,), generator=g, device='cuda', dtype=torch.uint8)
lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device='cuda')
seq = lut[d.long()] if G < 500_000_000 else torch.cat([lut[d[i:i + (1 << 28)].long()] for i in range(0, G, 1 << 28)])
seq_np = seq.cpu().numpy(); print('gen+copy', round(time.time() - t, 1), flush=True)
slots = int(G / 0.6) | 1
t = time.time(); blob = api.build_slots(seq_np, slots); print('build', G, round(time.time() - t, 1), flush=True)
