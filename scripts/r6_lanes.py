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
    if mode == "idle":  # does what else the process holds on the device slow the lanes?  (bench.py's null sink runs 25 % under this script's on the same box)
        def null_runs(tag):
            reps = [api.map_files(index, fq, samout=os.path.join(d, "x.sam"), first_gpu=0, gpus=1, streams=2, discard_sam=True, cmdline="lanes") for _ in range(4)]
            r = reps[-1]
            print(f"{tag}: {[round(x['reads'] / x['seconds'] / 1e6, 2) for x in reps]} M reads/s; last: wall {r['seconds']:.3f} s, lane busy {r['gpu_s']:.3f}, map {r['dev_map_s']:.3f} parse {r['dev_parse_s']:.3f} "
                  f"format {r['dev_format_s']:.3f} h2d {r['dev_h2d_s']:.3f} d2h {r['dev_d2h_s']:.3f}", flush=True)
        null_runs("nothing else on the device")
        if len(sys.argv) > 4 and sys.argv[4] == "files_first":
            # bench.py's order: two calls into ONE FILE first, then the oracle on the head of the file, then the null sink
            for _ in range(2):
                r = api.map_files(index, fq, samout=os.path.join(d, "one.sam"), first_gpu=0, gpus=1, streams=2, cmdline="lanes")
                print(f"one file: {r['reads'] / r['seconds'] / 1e6:.2f} M reads/s", flush=True)
            null_runs("after two calls into one file")
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as ol
            oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
            head = os.path.join(d, "head.fq")
            with open(fq, "rb") as f, open(head, "wb") as g:
                g.write(f.read(400000 * 315))
            oi.map_file_se(head, os.path.join(d, "head.sam"), threads=bench.host_cores())
            null_runs("after the oracle mapped the head of the file (16 OpenMP threads)")
            import subprocess
            if os.path.exists(ol.REF_BIN):
                ufi = os.path.join(d, "idx.ufi")
                oi.save(ufi)
                subprocess.run([ol.REF_BIN, "-map", head, "-ufi", ufi, "-samout", os.path.join(d, "ref.sam"), "-threads", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
                os.remove(ufi)
                null_runs("after the reference binary ran on the head (index written to and read from /dev/shm)")
            settings = []
            raise SystemExit(0)
        d_seq2 = torch.from_numpy(seq_np).to(dev)
        ms = [api.Mapper(index, device=0) for _ in range(2)]
        wl = bench.Workload(torch, api, dev, d_seq2, lens, offs, False, 150, 0.01, 0.001, 1_000_000, 2, 4242, contexts=2)
        wl.timed(ms, 4, 2)
        null_runs("two idle mapping contexts (8.5 GB of arrays each) + a 1 M-read workload resident")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
        wl.check(oi, 20000, bench.host_cores())
        null_runs("... and the oracle library loaded, its OpenMP team started")
        for m in ms:
            m.close()
        del wl, d_seq2
        torch.cuda.empty_cache()
        null_runs("contexts closed again")
        raise SystemExit(0)
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
