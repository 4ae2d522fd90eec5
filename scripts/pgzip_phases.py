#!/usr/bin/env python3
"""Host-side measurement (the GPU box's CPUs): urmapx_gunzip_file on a gzip -1 stream of bench.py's FASTQ shape (2 M reads of 150 bases, quality 'I'),
the rounds' phases printed by URMAPX_PGZIP_VERBOSE.  usage: pgzip_phases.py [reads] [threads...]"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from urmap_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
threads = [int(x) for x in sys.argv[2:]] or [16]
d = "/dev/shm/pgzip_phases"
os.makedirs(d, exist_ok=True)
fq = os.path.join(d, "r.fq")
rng = np.random.default_rng(3)
reads = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n * 150)]
bench.write_fastq_fixed(fq, reads, n, 150)
with open(fq + ".gz", "wb") as f:
    subprocess.run(["gzip", "-1", "-c", fq], stdout=f, check=True)
print("fastq", os.path.getsize(fq), "gz", os.path.getsize(fq + ".gz"), flush=True)
os.environ["URMAPX_PGZIP_VERBOSE"] = "1"
for th in threads:
    for rep in range(2):
        t = time.time()
        st = api.gunzip_file(fq + ".gz", fq + ".out", th)
        dt = time.time() - t
        print(f"threads {th}: {st} {dt:.3f} s {st[0] / dt / 1e9:.2f} GB/s", flush=True)
t = time.time(); subprocess.run(["gzip", "-dc", fq + ".gz"], stdout=subprocess.DEVNULL, check=True); print(f"gzip -dc: {os.path.getsize(fq) / (time.time() - t) / 1e9:.2f} GB/s")
import shutil; shutil.rmtree(d)
