#!/usr/bin/env python3
"""BASELINE config 4's input at its own size on ONE device: 100 M x 150 bp reads -- 31.5 GB of FASTQ in
/dev/shm -> 34 GB of SAM -- against the hg38-scale index, through urmapx_map_files (= urmap -map, cmd_map of map.cpp:43-61)
  (1) into one SAM file,
  (2) into 8 shards (urmap -map -gpus 1 -samshards 8: what each of config 4's eight devices would run on an eighth of the file),
  (3) through the command line as a process of its own, index load included (URMAP_CONFIG4_CLI=1).
Checked: every record of the one file in input order (labels), `cat` of the shards == the one file, the first 100 k records, the
100 k around byte 2^32 of the SAM file, a middle 100 k and the LAST 100 k against the oracle.  Result: one JSON object (stdout and --out).

The oracle (tests/oracle_lib.py) is the checker here, as in bench.py's parity legs; nothing of the product path touches it."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

L = 150
REC = 11 + L + 3 + L + 1


def write_reads(f, reads, lo):
    n = reads.size // L
    a = np.empty((n, REC), dtype=np.uint8)
    a[:, 0] = ord("@"); a[:, 1] = ord("r")
    idx = np.arange(lo, lo + n, dtype=np.int64)
    for k in range(8):  # 8 digits hold 99 999 999: read 100 000 000 would need a ninth -- the labels count modulo 1e8 and say so
        a[:, 2 + k] = ((idx // 10 ** (7 - k)) % 10 + ord("0")).astype(np.uint8)
    a[:, 10] = ord("\n")
    a[:, 11:11 + L] = reads.reshape(n, L)
    a[:, 11 + L] = ord("\n"); a[:, 12 + L] = ord("+"); a[:, 13 + L] = ord("\n")
    a[:, 14 + L:14 + 2 * L] = ord("I")
    a[:, REC - 1] = ord("\n")
    a.tofile(f)


def record_starts(path):
    size = os.path.getsize(path)
    out = [np.zeros(1, np.int64)]
    piece = 512 << 20
    with open(path, "rb") as f:
        for lo in range(0, size, piece):
            b = np.frombuffer(f.read(piece), dtype=np.uint8)
            out.append(np.flatnonzero(b == 10).astype(np.int64) + (lo + 1))
    st = np.concatenate(out)
    assert st[-1] == size
    st = st[:-1]
    n_hdr = 0
    with open(path, "rb") as f:
        while f.read(1) == b"@":
            n_hdr += 1
            f.seek(int(st[n_hdr]))
    return st[n_hdr:], size


def labels_in_order(path, starts, size, n_reads):
    bad = 0
    with open(path, "rb") as f:
        step = 2_000_000
        for lo in range(0, n_reads, step):
            hi = min(n_reads, lo + step)
            a, b = int(starts[lo]), int(starts[hi]) if hi < n_reads else size
            f.seek(a)
            buf = np.frombuffer(f.read(b - a), dtype=np.uint8)
            rel = (starts[lo:hi] - a).astype(np.int64)
            num = np.zeros(hi - lo, np.int64)
            for k in range(8):
                num = num * 10 + (buf[rel + 1 + k].astype(np.int64) - ord("0"))
            bad += int((num != np.arange(lo, hi) % 100_000_000).sum()) + int((buf[rel] != ord("r")).sum())
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100_000_000)
    ap.add_argument("--mbp", type=float, default=3100)
    ap.add_argument("--out", default="gpurun_out/r6_config4/config4_one_device.json")
    ap.add_argument("--shards", type=int, default=8)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    import torch
    import bench
    import oracle_lib as ol
    from urmap_amd import api, ranks
    dev = torch.device("cuda", 0)
    R = ranks.Ranks().init(torch)
    t0 = time.time()
    d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(args.mbp * 1e6), dev)
    slots, fasta_bytes = bench.default_slot_count(lens, labels)
    index, blob_np, seq_np, d_seq, info = bench.place_index(R, torch, api, dev, d_seq, slots, lens, offs, labels)
    ok, vrep = index.validate()
    table_ck, genome_ck = index.checksum()
    oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, lens, offs, labels)
    cores = bench.host_cores()
    res = {"what": "BASELINE config 4's input (100 M x 150 bp SE reads) through urmapx_map_files on one MI355X", "reads": args.reads,
           "genome_checksum": f"{genome_ck:016x}", "slot_table_checksum": f"{table_ck:016x}", "slots": int(slots), "index_validated": bool(ok),
           "setup_s": round(time.time() - t0, 1), "host_cpus_granted": cores, "host_logical_cpus": os.cpu_count()}
    d = tempfile.mkdtemp(prefix="urmap_c4_", dir="/dev/shm")
    try:
        fq = os.path.join(d, "reads.fq")
        t0 = time.time()
        slab = 4_000_000
        with open(fq, "wb") as f:
            for lo in range(0, args.reads, slab):
                n = min(slab, args.reads - lo)
                r = bench.make_reads_torch(torch, 40_000 + lo // slab, d_seq, lens, offs, n, L, 0.01, 0.001, dev).cpu().numpy()
                write_reads(f, r, lo)
        res["fastq_bytes"] = os.path.getsize(fq)
        res["fastq_written_in_s"] = round(time.time() - t0, 1)
        assert res["fastq_bytes"] == args.reads * REC
        sam = os.path.join(d, "one.sam")
        keys = ("seconds", "parse_s", "gpu_s", "format_s", "write_s", "dev_h2d_s", "dev_parse_s", "dev_map_s", "dev_format_s", "dev_d2h_s", "shard_scan_s")
        runs = {}
        for name, kw in (("one_file", {}), ("shards", {"sam_shards": args.shards}), ("null_sink", {"discard_sam": True})):
            out = sam if name != "shards" else sam + ".sh"
            rep = bench.watched(lambda: api.map_files(index, fq, samout=out if name != "null_sink" else sam + ".null", first_gpu=0, gpus=1, streams=2,
                                                      cmdline="r6_config4", **kw))
            runs[name] = {"reads": int(rep["reads"]), "reads_per_s": round(rep["reads"] / rep["seconds"], 1), "lanes": rep["lanes"], "shards": rep["shards"],
                          "medium": rep["medium"].decode(), "placement": rep["placement"].decode(), "host": rep["host"],
                          "mapped_q10_frac": round(rep["mapped_q"] / max(1, rep["reads"]), 4), **{k: round(rep[k], 3) for k in keys}}
            assert rep["reads"] == args.reads, (name, rep["reads"])
        res["runs"] = runs
        res["sam_bytes"] = os.path.getsize(sam)
        parts = [sam + f".sh.{k}" for k in range(args.shards)]
        res["shard_bytes"] = [os.path.getsize(p) for p in parts]
        t0 = time.time()
        res["cat_of_shards_equals_the_one_file"] = bool(bench.files_equal_concat(sam, parts))
        res["cat_compare_s"] = round(time.time() - t0, 1)
        for p in parts:
            os.remove(p)
        t0 = time.time()
        starts, size = record_starts(sam)
        res["records"] = int(len(starts))
        res["records_out_of_order_or_mislabelled"] = labels_in_order(sam, starts, size, args.reads) if len(starts) == args.reads else None
        res["order_scan_s"] = round(time.time() - t0, 1)
        n_chk = 100_000
        k_sam = int(np.searchsorted(starts, 2 ** 32))
        cuts = {"first": 0, "around_sam_byte_2^32": max(0, min(args.reads - n_chk, k_sam - n_chk // 2)), "middle": args.reads // 2, "last": args.reads - n_chk}
        slices = {}
        for name, lo in cuts.items():
            part, osam = os.path.join(d, "slice.fq"), os.path.join(d, "slice.oracle.sam")
            with open(fq, "rb") as f, open(part, "wb") as g:
                f.seek(lo * REC)
                g.write(f.read(n_chk * REC))
            oi.map_file_se(part, osam, threads=cores)
            want = [l for l in open(osam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
            a, b = int(starts[lo]), int(starts[lo + n_chk]) if lo + n_chk < len(starts) else size
            with open(sam, "rb") as f:
                f.seek(a)
                got = f.read(b - a).split(b"\n")[:-1]
            slices[name] = {"reads": [lo, lo + n_chk], "sam_bytes_from": a, "identical_to_oracle": bool(got == want and len(want) == n_chk)}
        res["slices_vs_oracle"] = slices
        os.remove(sam)
        if os.environ.get("URMAP_CONFIG4_CLI"):
            # the command line from start to exit, index load included: the .ufi in /dev/shm, this process's replica off the device first
            ufi = os.path.join(d, "idx.ufi")
            t0 = time.time()
            oi.save(ufi)
            res["ufi_written_in_s"] = round(time.time() - t0, 1)
            index.close()
            del d_seq
            torch.cuda.empty_cache()
            api.lib().urmapx_host_pool_trim()
            time.sleep(8)
            exe = os.path.join(ROOT, "urmap_amd", "urmap")
            cli = {}
            for name, extra in (("samshards_8", ["-samshards", str(args.shards)]), ("one_file", [])):
                t = time.time()
                r = subprocess.run([exe, "-map", fq, "-ufi", ufi, "-samout", sam, "-gpus", "1"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   env=dict(os.environ, URMAPX_VERBOSE="1"))
                wall = time.time() - t
                outs = [sam + f".{k}" for k in range(args.shards)] if extra else [sam]
                cli[name] = {"rc": r.returncode, "wall_s": round(wall, 2), "reads_per_s_of_wall": round(args.reads / wall, 1),
                             "sam_bytes": sum(os.path.getsize(p) for p in outs if os.path.exists(p)), "tail": (r.stdout + r.stderr).decode("latin-1")[-600:]}
                for p in outs:
                    if os.path.exists(p):
                        os.remove(p)
                time.sleep(8)
            res["cli"] = cli
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(res))
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
