#!/usr/bin/env python3
"""Diagnostic (GPU box): the file-to-file lanes with the SAM text dropped (urmapx_map_files, discard_sam) on 10 M reads, against how many
blocks per CU the persistent search kernel takes (URMAPX_BLOCKS_PER_CU) and how many lanes share the device: does a lane's chain of small
launches (phase 6's rounds, the text kernels, the copies) get onto the device while the other lane's search kernel runs?
usage: r6_lanes.py [genome_mbp] [n_reads]"""
import os
import sys
import tempfile
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import torch

import bench
from urmap_amd import api, ranks

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda", 0)
R = ranks.Ranks().init(torch)
d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(mbp * 1e6), dev)
slots, _ = bench.default_slot_count(lens, labels)
index, blob_np, seq_np, d_seq, tm = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
d = tempfile.mkdtemp(prefix="urmap_lanes_", dir="/dev/shm")
try:
    fq = os.path.join(d, "r.fq")
    with open(fq, "wb") as f:
        for lo in range(0, n, 2_000_000):
            k = min(2_000_000, n - lo)
            r = bench.make_reads_torch(torch, 777 + lo, d_seq, lens, offs, k, 150, 0.01, 0.001, dev).cpu().numpy()
            tmp = os.path.join(d, "part.fq")
            bench.write_fastq_fixed(tmp, r, k, 150)
            f.write(open(tmp, "rb").read())
            os.remove(tmp)
    del d_seq
    torch.cuda.empty_cache()
    mode = sys.argv[3] if len(sys.argv) > 3 else "blocks"
    if mode == "files":  # two and three lanes, SAM text dropped and into files
        settings = [(2, {}), (3, {}), (2, {}), (3, {})]
    elif mode == "ramp":  # the chunk-size ramp of the text phase against chunks of one size, alternating
        settings = [(2, {"URMAPX_NO_CHUNK_RAMP": "1"}), (2, {}), (2, {"URMAPX_NO_CHUNK_RAMP": "1"}), (2, {}), (3, {"URMAPX_NO_CHUNK_RAMP": "1"}), (3, {})]
    else:
        settings = [(st, {"URMAPX_BLOCKS_PER_CU": str(b)} if b else {}) for st in (2, 3) for b in (0, 15, 14, 12, 0)]
    for streams, env in settings:
        for k in ("URMAPX_NO_CHUNK_RAMP", "URMAPX_BLOCKS_PER_CU"):
            os.environ.pop(k, None)
        os.environ.update(env)
        reps = [api.map_files(index, fq, samout=os.path.join(d, "x.sam"), first_gpu=0, gpus=1, streams=streams, discard_sam=True, cmdline="lanes") for _ in range(4)]
        r = reps[-1]
        print(f"streams {streams} {env or 'default'}: {[round(x['reads'] / x['seconds'] / 1e6, 2) for x in reps]} M reads/s; last: wall {r['seconds']:.3f} s, "
              f"lane busy {r['gpu_s']:.3f}, stream time map {r['dev_map_s']:.3f} (search {r['dev_map_search_s']:.3f}, dp {r['dev_map_dp_s']:.3f}) parse {r['dev_parse_s']:.3f} "
              f"format {r['dev_format_s']:.3f} h2d {r['dev_h2d_s']:.3f} d2h {r['dev_d2h_s']:.3f}, alloc calls {r['alloc_dev_calls']}", flush=True)
    if mode in ("ramp", "files"):  # into a file too
        for env, st in (({"URMAPX_NO_CHUNK_RAMP": "1"}, 2), ({}, 2), ({"URMAPX_NO_CHUNK_RAMP": "1"}, 2), ({}, 2)) if mode == "ramp" else (({}, 2), ({}, 3), ({}, 2), ({}, 3)):
            os.environ.pop("URMAPX_NO_CHUNK_RAMP", None)
            os.environ.update(env)
            reps = [api.map_files(index, fq, samout=os.path.join(d, "x.sam"), first_gpu=0, gpus=1, streams=st, cmdline="lanes") for _ in range(3)]
            print(f"one file streams {st} {env or 'default'}: {[round(x['reads'] / x['seconds'] / 1e6, 2) for x in reps]} M reads/s, write_s {reps[-1]['write_s']:.3f}", flush=True)
            reps = [api.map_files(index, fq, samout=os.path.join(d, "y.sam"), first_gpu=0, gpus=1, streams=st, cmdline="lanes", sam_shards=2) for _ in range(3)]
            print(f"two shards streams {st} {env or 'default'}: {[round(x['reads'] / x['seconds'] / 1e6, 2) for x in reps]} M reads/s", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
