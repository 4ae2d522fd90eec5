#!/usr/bin/env python3
"""GPU box, one-off: is the hg38-scale index the product builds the reference's index?

  1. the bench's genome (bench.make_genome_torch, seed 20260101) is written as a FASTA file (60 bases per line, the layout
     bench.default_slot_count assumes), so that cmd_make_ufi's default slot count (GetPrime(file size / 0.6)) is the bench's;
  2. oracle/_ref/urmap (the UNMODIFIED reference, oracle/Makefile) -make_ufi   -> ref.ufi     (wall time = f1's CPU baseline)
  3. urmap_amd/urmap -make_ufi              (GPU counting passes + host inserts) -> gpu.ufi,  cmp with ref.ufi
  4. urmap_amd/urmap -make_ufi -host        (everything on the host)            -> host.ufi, cmp with ref.ufi
  5. urmap_amd/urmap -ufi_validate ref.ufi  (the device's UFIndex::Validate pass over the reference's own file)
Steps 2 and 3-4 run side by side when --parallel is given (the reference build is one thread).

usage: r5_ufi_fullscale.py [--mbp 3100] [--out gpurun_out/r5_ufi] [--dir /dev/shm/urmap_r5] [--parallel] [--skip-host] [--ref-validate]
Writes <out>/result.json after every step (a call cut short still leaves what it had)."""
import argparse
import hashlib
import json
import os
import resource
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.path.join(ROOT, "oracle", "_ref", "urmap")
EXE = os.path.join(ROOT, "urmap_amd", "urmap")


def write_fasta(args):
    """child process: genome on the GPU, FASTA file on the host"""
    import numpy as np
    import torch
    import bench
    dev = torch.device("cuda", 0)
    d_seq, lens, offs, labels, desc = bench.make_genome_torch(torch, 20260101, int(args.mbp * 1e6), dev)
    slots, fasta_bytes = bench.default_slot_count(lens, labels)
    seq = d_seq.cpu().numpy()
    W = 60
    with open(args.fasta, "wb") as f:
        for lab, L, off in zip(labels, lens, offs):
            L, off = int(L), int(off)
            f.write(b">" + lab.encode() + b"\n")
            s = seq[off:off + L]
            rows = L // W
            step = 1 << 22  # rows per piece
            for r0 in range(0, rows, step):
                r1 = min(rows, r0 + step)
                a = np.empty((r1 - r0, W + 1), np.uint8)
                a[:, :W] = s[r0 * W:r1 * W].reshape(r1 - r0, W)
                a[:, W] = 10
                f.write(a.tobytes())
            if L % W:
                f.write(s[rows * W:].tobytes() + b"\n")
    assert os.path.getsize(args.fasta) == fasta_bytes, (os.path.getsize(args.fasta), fasta_bytes)
    print(json.dumps({"slots": int(slots), "fasta_bytes": int(fasta_bytes), "genome": desc, "sequences": len(labels)}))


def sha_and_cmp(a, b=None, block=64 << 20):
    """sha256 of file a; with b: also whether the two files are equal byte for byte and the first differing offset"""
    h = hashlib.sha256()
    equal, first = True, None
    with open(a, "rb") as fa:
        fb = open(b, "rb") if b else None
        off = 0
        while True:
            x = fa.read(block)
            if not x:
                if fb and fb.read(1):
                    equal, first = False, off
                break
            h.update(x)
            if fb:
                y = fb.read(len(x))
                if equal and x != y:
                    equal = False
                    import numpy as np
                    n = min(len(x), len(y))
                    d = np.nonzero(np.frombuffer(x, np.uint8, n) != np.frombuffer(y, np.uint8, n))[0]
                    first = off + (int(d[0]) if len(d) else n)
            off += len(x)
        if fb:
            fb.close()
    return h.hexdigest(), equal, first


def timed(cmd, log):
    t0 = time.time()
    with open(log, "wb") as f:
        p = subprocess.Popen(cmd, stdout=f, stderr=subprocess.STDOUT)
    return p, t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbp", type=float, default=3100)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r5_ufi"))
    ap.add_argument("--dir", default="/dev/shm/urmap_r5")
    ap.add_argument("--parallel", action="store_true")
    ap.add_argument("--skip-host", action="store_true")
    ap.add_argument("--ref-validate", action="store_true", help="also time the reference's own -ufi_validate on ref.ufi (CPU, one thread)")
    ap.add_argument("--write-fasta", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--fasta", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.write_fasta:
        return write_fasta(args)
    os.makedirs(args.out, exist_ok=True)
    os.makedirs(args.dir, exist_ok=True)
    res = {"genome_mbp": args.mbp, "host": {"cpus_granted": len(os.sched_getaffinity(0)), "logical_cpus": os.cpu_count()}, "steps": {}}
    try:
        res["host"]["mem_total_GB"] = round(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9, 1)
    except (ValueError, OSError):
        pass

    def save():
        with open(os.path.join(args.out, "result.json"), "w") as f:
            json.dump(res, f, indent=1)
    fa = os.path.join(args.dir, "g.fa")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--write-fasta", "--mbp", str(args.mbp), "--fasta", fa], stdout=subprocess.PIPE, check=True)
    res["steps"]["fasta"] = {**json.loads(r.stdout.decode().strip().splitlines()[-1]), "seconds": round(time.time() - t0, 1)}
    save()
    ref_ufi, gpu_ufi, host_ufi = (os.path.join(args.dir, x) for x in ("ref.ufi", "gpu.ufi", "host.ufi"))

    def finish_ref(p, t0):
        rc = p.wait()
        dt = time.time() - t0
        ru = resource.getrusage(resource.RUSAGE_CHILDREN)
        res["steps"]["reference_make_ufi"] = {"cmd": "oracle/_ref/urmap -make_ufi g.fa -output ref.ufi", "rc": rc, "wall_s": round(dt, 1), "threads": 1,
                                               "ufi_bytes": os.path.getsize(ref_ufi) if os.path.exists(ref_ufi) else 0,
                                               "children_maxrss_GB": round(ru.ru_maxrss / 1e6, 1)}
        save()
        return rc

    pref, tref = timed([REF, "-make_ufi", fa, "-output", ref_ufi], os.path.join(args.out, "ref_make_ufi.log"))
    if not args.parallel and finish_ref(pref, tref) != 0:
        return 1
    builds = [("product_gpu", gpu_ufi, [EXE, "-make_ufi", fa, "-output", gpu_ufi, "-quiet"])]
    if not args.skip_host:
        builds.append(("product_host", host_ufi, [EXE, "-make_ufi", fa, "-output", host_ufi, "-host", "-quiet"]))
    shas = {}
    for name, path, cmd in builds:
        p, t0 = timed(cmd, os.path.join(args.out, name + ".log"))
        rc = p.wait()
        res["steps"][name] = {"cmd": " ".join(os.path.relpath(c, ROOT) if c.startswith(ROOT) else os.path.basename(c) for c in cmd), "rc": rc,
                              "wall_s": round(time.time() - t0, 1), "ufi_bytes": os.path.getsize(path) if os.path.exists(path) else 0,
                              "beside_the_reference_build": bool(args.parallel and pref.poll() is None)}
        save()
        if rc != 0:
            return 1
        if args.parallel:  # compared once the reference's file exists; keep only the hash and the file until then
            continue
    if args.parallel and finish_ref(pref, tref) != 0:
        return 1
    t0 = time.time()
    ref_sha = None
    for name, path, _ in builds:
        sha, equal, first = sha_and_cmp(path, ref_ufi)
        shas[name] = sha
        if ref_sha is None:
            ref_sha = sha_and_cmp(ref_ufi)[0]
        res["steps"][name].update({"sha256": sha, "cmp_with_reference": "equal" if equal else f"DIFFERS at byte {first}"})
        os.remove(path)
        save()
    res["steps"]["reference_make_ufi"]["sha256"] = ref_sha
    res["compare_s"] = round(time.time() - t0, 1)
    save()
    # the device's Validate pass over the REFERENCE's file
    p, t0 = timed([EXE, "-ufi_validate", ref_ufi], os.path.join(args.out, "validate_ref_ufi.log"))
    rc = p.wait()
    res["steps"]["product_ufi_validate_on_ref_ufi"] = {"rc": rc, "wall_s": round(time.time() - t0, 1),
                                                       "log": open(os.path.join(args.out, "validate_ref_ufi.log"), "rb").read().decode("latin-1")[-400:]}
    save()
    if args.ref_validate:
        p, t0 = timed([REF, "-ufi_validate", ref_ufi], os.path.join(args.out, "ref_validate.log"))
        try:
            rc = p.wait(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill()
            rc = "timeout after 1500 s"
        res["steps"]["reference_ufi_validate"] = {"rc": rc, "wall_s": round(time.time() - t0, 1), "threads": 1}
        save()
    res["all_equal"] = all(res["steps"][n].get("cmp_with_reference") == "equal" for n, _, _ in builds)
    save()
    shutil.rmtree(args.dir, ignore_errors=True)
    print(json.dumps(res))
    return 0 if res["all_equal"] else 1


if __name__ == "__main__":
    sys.exit(main())
