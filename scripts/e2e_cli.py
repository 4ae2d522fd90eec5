#!/usr/bin/env python3
"""End-to-end timing of the `urmap` command line on the GPU box: FASTQ file in -> SAM file out.

Generates a synthetic genome (bench.py's generator), writes FASTA + FASTQ under /tmp, builds the .ufi with
`urmap -make_ufi`, then times `urmap -map` (wall clock of the whole process: index load + upload included, and
reported separately from the steady-state rate the tool prints).  Diagnostic, not the headline metric.
"""
import argparse, os, subprocess, sys, time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def write_fasta(path, seq_np, seq_lengths, seq_offsets, labels, width=80):
    with open(path, "wb") as f:
        for lab, L, o in zip(labels, seq_lengths, seq_offsets):
            f.write(b">" + lab.encode() + b"\n")
            s = seq_np[o:o + L]
            full = (L // width) * width
            body = np.empty((L // width, width + 1), dtype=np.uint8)
            body[:, :width] = s[:full].reshape(-1, width)
            body[:, width] = 10
            f.write(body.tobytes())
            if full < L:
                f.write(s[full:].tobytes() + b"\n")


def write_fastq(path, reads_np, n, L, tag="r"):
    lab = np.array([f"@{tag}{i:08d}\n".encode() for i in range(n)], dtype=f"S{len(tag) + 10}")
    labw = lab.dtype.itemsize
    rec = np.empty((n, labw + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :labw] = np.frombuffer(lab.tobytes(), dtype=np.uint8).reshape(n, labw)
    rec[:, labw:labw + L] = reads_np.reshape(n, L)
    rec[:, labw + L:labw + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, labw + L + 3:labw + 2 * L + 3] = ord("I")
    rec[:, -1] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=400)
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--dir", default="/tmp/urmap_e2e")
    ap.add_argument("--pe", action="store_true")
    ap.add_argument("--extra", default="")
    args = ap.parse_args()
    import torch
    os.makedirs(args.dir, exist_ok=True)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "urmap_amd", "urmap")
    dev = torch.device("cuda", 0)
    L = args.read_len
    t0 = time.time()
    d_seq, seq_lengths, seq_offsets, labels = bench.make_genome_torch(torch, 20260101, int(args.genome_mbp * 1e6), dev)
    seq_np = d_seq.cpu().numpy()
    fa, ufi = os.path.join(args.dir, "g.fa"), os.path.join(args.dir, "g.ufi")
    write_fasta(fa, seq_np, seq_lengths, seq_offsets, labels)
    slots = bench.next_prime(int(os.path.getsize(fa) / 0.6))
    print(f"genome + FASTA: {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    subprocess.check_call([exe, "-make_ufi", fa, "-output", ufi, "-slots", str(slots)])
    print(f"make_ufi: {time.time() - t0:.1f} s ({os.path.getsize(ufi) / 1e9:.2f} GB)", flush=True)
    t0 = time.time()
    fq1, fq2, sam = os.path.join(args.dir, "r1.fq"), os.path.join(args.dir, "r2.fq"), os.path.join(args.dir, "out.sam")
    if args.pe:
        npairs = args.reads // 2
        r = bench.make_pairs_torch(torch, 7, d_seq, seq_lengths, seq_offsets, npairs, L, 0.01, 0.015, dev).cpu().numpy()
        r = r.reshape(npairs, 2, L)
        write_fastq(fq1, np.ascontiguousarray(r[:, 0]), npairs, L, "p")
        write_fastq(fq2, np.ascontiguousarray(r[:, 1]), npairs, L, "p")
        cmd = [exe, "-map2", fq1, "-reverse", fq2, "-ufi", ufi, "-samout", sam]
        nreads = 2 * npairs
    else:
        r = bench.make_reads_torch(torch, 7, d_seq, seq_lengths, seq_offsets, args.reads, L, 0.01, 0.001, dev).cpu().numpy()
        write_fastq(fq1, r, args.reads, L)
        cmd = [exe, "-map", fq1, "-ufi", ufi, "-samout", sam]
        nreads = args.reads
    cmd += args.extra.split()
    del d_seq
    torch.cuda.empty_cache()
    print(f"FASTQ: {time.time() - t0:.1f} s", flush=True)
    t0 = time.time()
    env = dict(os.environ, URMAPX_VERBOSE="1")
    subprocess.check_call(cmd, env=env)
    dt = time.time() - t0
    print(f"urmap {'-map2' if args.pe else '-map'}: {nreads} reads in {dt:.2f} s wall = {nreads / dt / 1e6:.2f} M reads/s "
          f"(FASTQ {os.path.getsize(fq1) / 1e9:.2f} GB -> SAM {os.path.getsize(sam) / 1e9:.2f} GB)", flush=True)


if __name__ == "__main__":
    main()
