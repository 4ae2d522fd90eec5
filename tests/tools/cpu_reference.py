#!/usr/bin/env python3
"""Test-side tool (it runs the oracle and the compiled reference, so it lives under tests/).
CPU side of the comparison on the GPU box's host: the reference binary itself (oracle/_ref/urmap, built from
/root/reference/src by oracle/Makefile; it travels with the repo snapshot) and the oracle port, timed on the same
FASTQ + .ufi files the product CLI maps.  Prints reads/s (index load excluded for all three).  Diagnostic script."""
import argparse, os, re, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import bench  # noqa: E402
import e2e_cli  # noqa: E402


def effective_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=400)
    ap.add_argument("--reads", type=int, default=2_000_000)
    ap.add_argument("--dir", default="/tmp/urmap_cpuref")
    args = ap.parse_args()
    import torch
    os.makedirs(args.dir, exist_ok=True)
    dev = torch.device("cuda", 0)
    L = 150
    d_seq, seq_lengths, seq_offsets, labels, _desc = bench.make_genome_torch(torch, 20260101, int(args.genome_mbp * 1e6), dev)
    seq_np = d_seq.cpu().numpy()
    fa, ufi, fq = (os.path.join(args.dir, x) for x in ("g.fa", "g.ufi", "r.fq"))
    e2e_cli.write_fasta(fa, seq_np, seq_lengths, seq_offsets, labels)
    slots = bench.next_prime(int(os.path.getsize(fa) / 0.6))
    exe = os.path.join(ROOT, "urmap_amd", "urmap")
    subprocess.check_call([exe, "-make_ufi", fa, "-output", ufi, "-slots", str(slots)])
    r = bench.make_reads_torch(torch, 7, d_seq, seq_lengths, seq_offsets, args.reads, L, 0.01, 0.001, dev).cpu().numpy()
    e2e_cli.write_fastq(fq, r, args.reads, L)
    ncpu = effective_cpus()
    print(f"host: {os.cpu_count()} logical CPUs, {ncpu} usable (affinity / cgroup quota)", flush=True)
    ref = os.path.join(ROOT, "oracle", "_ref", "urmap")
    sams = {}
    for threads in (ncpu, 10):
        out = os.path.join(args.dir, f"ref_{threads}.sam")
        t0 = time.time()
        p = subprocess.run([ref, "-map", fq, "-ufi", ufi, "-samout", out, "-threads", str(threads)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        wall = time.time() - t0
        txt = (p.stdout + p.stderr).decode(errors="replace")
        m = re.search(r"(\d+)\s+Seconds to load index", txt)
        m1 = re.search(r"(\d+)\s+Seconds in mapper", txt)
        m2 = re.search(r"(\d+)\s+Reads/sec", txt)
        load = float(m.group(1)) if m else float("nan")
        print(f"reference urmap -threads {threads}: wall {wall:.1f} s; its own report (whole seconds): load {load:.0f} s, "
              f"mapper {m1.group(1) if m1 else '?'} s, {m2.group(1) if m2 else '?'} reads/s; reads / (wall - load) = "
              f"{args.reads / max(1e-9, wall - load):.0f} reads/s", flush=True)
        sams[threads] = out
    import oracle_lib as ol
    idx = ol.Index.load(ufi)
    for threads in (ncpu, os.cpu_count()):
        out = os.path.join(args.dir, f"port_{threads}.sam")
        t0 = time.time()
        idx.map_file_se(fq, out, threads=threads)
        dt = time.time() - t0
        print(f"oracle port, {threads} threads: {args.reads / dt:.0f} reads/s (FASTQ -> SAM, index already loaded)", flush=True)
    same = ol.sam_records(sams[ncpu]) == ol.sam_records(out)
    print(f"reference SAM == port SAM (records, @PG excluded; order-insensitive): "
          f"{sorted(ol.sam_records(sams[ncpu])) == sorted(ol.sam_records(out))} (same order: {same})", flush=True)
    t0 = time.time()
    gsam = os.path.join(args.dir, "gpu.sam")
    p = subprocess.run([exe, "-map", fq, "-ufi", ufi, "-samout", gsam], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    txt = p.stderr.decode(errors="replace")
    m2 = re.search(r"([0-9.]+)\s+Reads/sec", txt)
    print(f"product CLI on the GPU: {m2.group(1) if m2 else '?'} reads/s in the mapper; SAM == reference SAM: "
          f"{sorted(ol.sam_records(gsam)) == sorted(ol.sam_records(sams[ncpu]))}", flush=True)


if __name__ == "__main__":
    main()
