#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X mapping path (150 bp single-end, urmap -map) on synthetic data.

A "step" is one pass of the hot path (seed+probe kernel, then search/extend kernel) over one batch of
reads that is already resident in HBM, against an index that is resident in HBM.  One process per GPU;
the index is replicated, reads are sharded by rank, there is no collective on the data path.

`python bench.py --gpus N` with no launcher around it starts its own N ranks (child processes through
torch.distributed.run, before this process touches the GPU); under the driver's launcher it is one of the ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`, plus
`other_workloads` (paired-end and 250 bp single-end on the same resident index) and `e2e` (FASTQ file -> SAM file).
"""
from __future__ import annotations

import argparse
import json
import os
import warnings
import sys
import time

# Idle OpenMP workers sleep instead of spinning, as the `urmap` command line sets it (urmap_main.cpp): this process is granted 16 CPUs on the
# GPU boxes (a cgroup quota) and runs several OpenMP teams -- the oracle's, the pipeline threads' of the file-to-file legs -- whose spinning
# workers get the whole process throttled (round 5: the file-to-file legs ran at a third of their rate in half of the runs until the
# library made its own teams sleep; this covers the teams it does not own).  Before any OpenMP runtime is loaded.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
# The HIP runtime spreads a process's streams over four hardware queues by default, and streams that share a queue take turns.  This process has torch's stream, two mapping
# contexts' and, in the file-to-file legs, two streams per lane: the lanes ran at 29 M reads/s with the default and at 39-40 M with 8 or 16 queues (profiles/r6/hw_queues.txt).
# The command line and the library set the same (urmap_main.cpp, urmapx.hip); here torch starts the runtime first, so it is set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CODE_VERSION = "r6d"   # committed PMC profiles carry the code version they were taken on (pmc_traffic); r6b: two chunks of k-mer starts, row stores in LDS, PosToCoordL by lanes; r6c: + the instruction trims of calls 31-34 (profiles/r6/ab_instruction_trims.txt); r6d: + call 35

# work per read of the reference on the survey's 40 Mbp planning genome (instrumented build)
SURVEY_WORK_PER_READ = {
    "se150": {"n_getblob": 207, "n_rowcalls": 110, "n_rowhop": 229, "n_extend": 187, "n_extbases": 5228, "n_alignhsp": 0.66,
              "n_viterbi": 0.59, "n_dpcells": 1296},
    "se250": {"n_getblob": 454, "n_rowcalls": 170, "n_rowhop": 266, "n_extend": 373, "n_extbases": 13153, "n_alignhsp": 3.5,
              "n_viterbi": 4.3, "n_dpcells": 18416},
    "pe150": {"n_getblob": 183, "n_rowcalls": 61, "n_rowhop": 141, "n_extend": 124, "n_extbases": 5169, "n_viterbi": 0.37,
              "n_dpcells": 581},
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("URMAP_BENCH_GENOME_MBP", 3100)))
    ap.add_argument("--reads-per-step", type=int, default=int(os.environ.get("URMAP_BENCH_READS", 1_000_000)))
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--sub", type=float, default=0.01)
    ap.add_argument("--indel", type=float, default=0.001)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the PE and 250 bp runs on the same index")
    ap.add_argument("--no-e2e", action="store_true", help="skip the FASTQ file -> SAM file run")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("URMAP_BENCH_STREAMS", 1)),
                    help="mapping contexts (HIP streams) a batch is split over, as urmap -streams does; 1 keeps the launches of a step back to back so that their times add up to the step (2: -2 %% at 150 bp, +5 %% at 250 bp since round 4 made phase 6 cheap; launches overlap)")
    ap.add_argument("--contexts", type=int, default=int(os.environ.get("URMAP_BENCH_CONTEXTS", 2)),
                    help="mapping contexts of the device that whole batches alternate over, each on its own HIP stream -- how urmap -streams K (default 2) runs a device: "
                         "one context's search kernel fills the CUs that the other's tail and phase-6 launches leave idle (round 6: 48.9 -> 54 M reads/s at 150 bp, pairs 50.4 -> 63.2 M, "
                         "profiles/r6/two_contexts_and_batch_size.txt); 1 = one context, its steps back to back (rounds 1-5; --streams then splits a batch)")
    ap.add_argument("--mode", choices=("se", "pe"), default="se", help="pe: 2x150 read pairs through State2::Search4 (config 3; not the headline metric)")
    return ap.parse_args()


def _is_prime(n):
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in small:  # deterministic Miller-Rabin for 64-bit integers
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def next_prime(n):
    while not _is_prime(n):
        n += 1
    return n


def get_prime(n):
    """GetPrime (prime.cpp:11-21): first rung >= n of the reference's ladder of 410 primes; rung k is the first prime >=
    t_k, t_0 = 100, t_(k+1) = trunc(t_k / 0.95) in double arithmetic (same rule the command line uses, urmap_main.cpp)."""
    t = 100
    for _ in range(410):
        p = next_prime(t)
        if p >= n:
            return p
        t = int(float(t) / 0.95)
    raise ValueError("GetPrime overflow")


def default_slot_count(seq_lengths, labels, width=60):
    """cmd_make_ufi's default (ufindexio.cpp:138-150): GetPrime(FASTA file size in bytes / load factor 0.6), for the
    FASTA file this genome would be (">label\n" + lines of `width` bases)."""
    size = 0
    for L, lab in zip(seq_lengths, labels):
        L = int(L)
        size += 1 + len(lab) + 1 + L + (L + width - 1) // width
    return get_prime(int(float(size) / 0.6)), size


HG38_LENGTHS_MBP = [248.96, 242.19, 198.30, 190.21, 181.54, 170.81, 159.35, 145.14, 138.39, 133.80, 135.09, 133.28,
                    114.36, 107.04, 101.99, 90.34, 83.26, 80.37, 58.62, 64.44, 46.71, 50.82, 156.04, 57.23]


_M64 = (1 << 64) - 1


def _s64(x):
    """a 64-bit pattern as the signed scalar torch's int64 arithmetic takes"""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def _mix_int(x):
    """splitmix64's finaliser on a Python integer (the key schedule of the generator below)"""
    x &= _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def _mix_t(torch, x):
    """the same finaliser on an int64 tensor: multiplications wrap, `>>` is arithmetic on int64, so the vacated bits are masked off"""
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _s64(0x94D049BB133111EB)
    return x ^ ((x >> 31) & ((1 << 33) - 1))


def _stream_key(seed, a, b=0):
    """key of the random stream (a, b) of generator `seed`: stream values are _mix_t(key + counter)"""
    return _s64(_mix_int(_mix_int(seed) ^ _mix_int((a << 20) + b + 0x632BE59BD9B4E019)))


def make_genome_torch(torch, seed, total_bp, device, repeat_frac=0.5, n_frac=0.05, max_div=0.20):
    """Concatenated upper-case sequence store as -make_ufi lays it out (ufindex.cpp:462-511): 24 sequences with
    hg38's chromosome length proportions, joined by 32 '-' bytes.  SURVEY.md 8(d) input 2: ~`repeat_frac` of the bases
    are copies of repeat families whose copy numbers are log-uniform in 10..1e5 (family length 300..6000, each copy
    0..`max_div` diverged from the consensus, either strand), `n_frac` of the bases lie in runs of N.

    Round 6: the store is a pure function of (seed, total_bp) -- on any device, any box, any torch.  Every random value is a
    counter-based hash (splitmix64's finaliser over key + counter, integer arithmetic only) instead of a draw from the
    device's generator, and copies that overlap are laid down in a DEFINED order: families in sequence, the slabs of a family
    in sequence, the copies of a slab sorted by start with the later start winning (each copy writes up to the start of the
    next, so one index-put never names a byte twice; rounds 1-5 scattered overlapping copies in one index-put whose order of
    writes to a duplicate index is undefined, and every run mapped a slightly different genome).  tests/test_genome_cpu.py pins
    the 40 Mbp store's checksum on the CPU; the same value must come out on the GPU.
    Returns the device uint8 tensor, the directory and what was laid down."""
    scale = total_bp / (sum(HG38_LENGTHS_MBP) * 1e6)
    lens = [max(2000, int(x * 1e6 * scale)) for x in HG38_LENGTHS_MBP]
    offsets, off = [], 0
    for i, L in enumerate(lens):
        offsets.append(off)
        off += L + (32 if i + 1 != len(lens) else 0)
    size = off
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.tensor(list(b"TGCA"), dtype=torch.uint8, device=device)
    seq = torch.empty(size, dtype=torch.uint8, device=device)
    step = 1 << 27
    k_base = _stream_key(seed, 1)
    for lo in range(0, size, step):
        hi = min(size, lo + step)
        seq[lo:hi] = lut[(_mix_t(torch, torch.arange(lo, hi, device=device, dtype=torch.int64) + k_base) >> 40) & 3]
    lens_t = torch.tensor(lens, dtype=torch.int64, device=device)
    offs_t = torch.tensor(offsets, dtype=torch.int64, device=device)
    cum = torch.cumsum(lens_t, 0)
    total = int(sum(lens))
    budget = int(total_bp * repeat_frac)
    div_q = int(max_div * (1 << 24))
    laid, fams = 0, []
    hostg = np.random.Generator(np.random.PCG64(seed))
    while laid < budget and len(fams) < 4000:
        fl = int(hostg.integers(300, 6000))
        copies = int(round(10 ** hostg.uniform(1.0, 5.0)))
        copies = max(1, min(copies, (budget - laid) // fl + 1, max(10, int(0.2 * budget) // fl)))
        f = len(fams)
        ar = torch.arange(fl, device=device, dtype=torch.int64)
        fam_idx = (_mix_t(torch, ar + _stream_key(seed, 2, f)) >> 40) & 3
        k_copy, k_cell = _stream_key(seed, 3, f), _stream_key(seed, 4, f)
        done = 0
        while done < copies:  # copies of one family, a slab at a time
            n = min(copies - done, max(1, (1 << 25) // fl))
            c = torch.arange(done, done + n, device=device, dtype=torch.int64)
            hc = _mix_t(torch, c + k_copy)                       # per copy: bits 1..62 its place, 0..23 its divergence, 24 its strand
            u = ((hc >> 1) & ((1 << 62) - 1)) % total
            si = torch.searchsorted(cum, u, right=True).clamp(max=len(lens) - 1)
            p = u - (cum[si] - lens_t[si])
            p = torch.minimum(p, (lens_t[si] - fl - 1).clamp(min=0))
            ok = lens_t[si] > fl + 1
            start = (offs_t[si] + p)[ok]
            c, hc = c[ok], hc[ok]
            m = int(start.numel())
            done += n
            if m == 0:
                continue
            start, order = torch.sort(start, stable=True)
            c, hc = c[order], hc[order]
            nxt = torch.cat([start[1:], start[-1:] + fl])
            keep_len = torch.minimum(nxt - start, torch.full_like(start, fl))  # a copy is written up to where the next one starts
            thr = (((hc & 0xFFFFFF) * div_q) >> 24)[:, None]                    # this copy's divergence, 0..max_div, as a 24-bit threshold
            hb = _mix_t(torch, c[:, None] * fl + ar[None, :] + k_cell)          # per base: bits 0..23 "mutated?", 40..41 the random base
            idx4 = torch.where((hb & 0xFFFFFF) < thr, (hb >> 40) & 3, fam_idx[None, :].expand(m, fl))
            minus = ((hc >> 24) & 1).bool()[:, None]
            copy = torch.where(minus, comp[idx4.flip(1)], lut[idx4])
            keep = ar[None, :] < keep_len[:, None]
            pos = (start[:, None] + ar[None, :])[keep]
            seq[pos] = copy[keep]
            del thr, hb, idx4, copy, pos, keep
        laid += copies * fl
        fams.append((fl, copies))
    # N runs
    n_left = int(total_bp * n_frac)
    n_draw = max(1, n_left // 25000 + 8)
    rl_all = hostg.integers(100, 50000, n_draw).tolist()
    pos_u = hostg.random(n_draw).tolist()
    for rl, u in zip(rl_all, pos_u):
        if n_left <= 0:
            break
        rl = min(rl, n_left)
        k = int(u * len(lens)) % len(lens)
        if lens[k] <= rl + 2:
            continue
        p = offsets[k] + int(u * 1e9) % (lens[k] - rl)
        seq[p:p + rl] = ord("N")
        n_left -= rl
    for i in range(len(lens) - 1):
        seq[offsets[i] + lens[i]: offsets[i] + lens[i] + 32] = ord("-")
    labels = [f"chr{i + 1}" for i in range(22)] + ["chrX", "chrY"]
    cn = sorted(c for _, c in fams)
    desc = {"generator": "bench.make_genome_torch round 6 (counter-based, device-independent)", "seed": int(seed),
            "repeat_families": len(fams), "repeat_bases_laid": int(laid), "copy_number_min": cn[0] if cn else 0,
            "copy_number_median": cn[len(cn) // 2] if cn else 0, "copy_number_max": cn[-1] if cn else 0,
            "max_divergence": max_div, "n_frac": n_frac}
    return seq, np.array(lens, np.uint32), np.array(offsets, np.uint32), labels, desc


def array_checksum(a):
    """urmapx_checksum_device's value for a host byte array (numpy restatement: include/urmapx.h): the sum modulo 2^64 over the
    little-endian 64-bit words w_i, the last zero-padded, of murmur64(w_i + (i + 1) * 0x9E3779B97F4A7C15)"""
    a = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    n = a.size
    total = 0
    step = 1 << 24
    with np.errstate(over="ignore"):
        for lo in range(0, n, step):
            part = a[lo:lo + step]
            if part.size & 7:
                part = np.concatenate([part, np.zeros(8 - (part.size & 7), np.uint8)])
            w = part.view("<u8").astype(np.uint64)
            i = np.arange(lo // 8 + 1, lo // 8 + 1 + w.size, dtype=np.uint64)
            h = w + i * np.uint64(0x9E3779B97F4A7C15)
            h ^= h >> np.uint64(33)
            h *= np.uint64(0xFF51AFD7ED558CCD)
            h ^= h >> np.uint64(33)
            h *= np.uint64(0xC4CEB9FE1A85EC53)
            h ^= h >> np.uint64(33)
            total = (total + int(h.sum(dtype=np.uint64))) & _M64
    return total


def _read_starts(torch, g, d_seq, seq_lengths, seq_offsets, n, span, device):
    """n window starts, uniform over the sequences; a window that holds an N (an assembly gap: no sequencer reads come
    from there) is drawn again once."""
    ns = len(seq_lengths)
    lens_all = torch.tensor(seq_lengths.astype(np.int64), device=device)
    offs_all = torch.tensor(seq_offsets.astype(np.int64), device=device)

    def draw(k):
        si = torch.randint(0, ns, (k,), generator=g, device=device)
        return offs_all[si] + (torch.rand(k, generator=g, device=device, dtype=torch.float64) * (lens_all[si] - span - 2).clamp(min=1).double()).long()
    start = draw(n)
    probe = torch.tensor([0, span // 2, span - 1], device=device)
    bad = (d_seq[(start[:, None] + probe[None, :]).reshape(-1)].reshape(n, 3) == ord("N")).any(1)
    nb = int(bad.sum())
    if nb:
        start[bad] = draw(nb)
    return start


def make_reads_torch(torch, seed, d_seq, seq_lengths, seq_offsets, n, L, sub, indel, device):
    """n reads of length L sampled from the device-resident genome: substitutions, <=1 indel per read with
    probability indel*L, half reverse-complemented.  Returns uint8 tensor [n*L]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    start = _read_starts(torch, g, d_seq, seq_lengths, seq_offsets, n, L + 2, device)
    ar = torch.arange(L, device=device)
    # one indel per affected read: deletion (skip a base) or insertion (repeat index, then randomise the base)
    u = torch.rand(n, generator=g, device=device)
    has_del = u < (indel * L / 2)
    has_ins = (u >= indel * L / 2) & (u < indel * L)
    ipos = torch.randint(5, L - 5, (n,), generator=g, device=device)
    idx = ar[None, :].expand(n, L).clone()
    idx = idx + ((ar[None, :] >= ipos[:, None]) & has_del[:, None]).long()
    idx = idx - ((ar[None, :] > ipos[:, None]) & has_ins[:, None]).long()
    reads = d_seq[(start[:, None] + idx).reshape(-1)].reshape(n, L)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    rnd = acgt[torch.randint(0, 4, (n, L), generator=g, device=device)]
    m = torch.rand(n, L, generator=g, device=device) < sub
    m = m | ((ar[None, :] == ipos[:, None]) & has_ins[:, None])
    isbase = (reads == 65) | (reads == 67) | (reads == 71) | (reads == 84)
    reads = torch.where(m & isbase, rnd, reads)
    # reverse complement half of them
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    minus = torch.rand(n, generator=g, device=device) < 0.5
    rc = comp[reads.flip(1).long()]
    reads = torch.where(minus[:, None], rc, reads)
    return reads.reshape(-1).contiguous()


def make_pairs_torch(torch, seed, d_seq, seq_lengths, seq_offsets, npairs, L, sub1, sub2, device):
    """npairs FR pairs (insert ~ N(300, 50) clipped to [L+20, 600]), mates interleaved: uint8 tensor [2*npairs*L]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    ins = (300 + 50 * torch.randn(npairs, generator=g, device=device)).long().clamp(L + 20, 600)
    start = _read_starts(torch, g, d_seq, seq_lengths, seq_offsets, npairs, 700, device)
    ar = torch.arange(L, device=device)
    a = d_seq[(start[:, None] + ar[None, :]).reshape(-1)].reshape(npairs, L)
    b = d_seq[((start + ins - L)[:, None] + ar[None, :]).reshape(-1)].reshape(npairs, L)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for x, y in zip(b"ACGTN", b"TGCAN"):
        comp[x] = y

    def mutate(r, sub):
        rnd = acgt[torch.randint(0, 4, r.shape, generator=g, device=device)]
        m = torch.rand(r.shape, generator=g, device=device) < sub
        isbase = (r == 65) | (r == 67) | (r == 71) | (r == 84)
        return torch.where(m & isbase, rnd, r)
    a = mutate(a, sub1)
    b = comp[mutate(b, sub2).flip(1).long()]
    swap = torch.rand(npairs, generator=g, device=device) < 0.5
    r1 = torch.where(swap[:, None], b, a)
    r2 = torch.where(swap[:, None], a, b)
    return torch.stack([r1, r2], dim=1).reshape(-1).contiguous()


def pmc_traffic(kernel, reads_per_launch, total_bp, read_len=150, mode="se", code_version=CODE_VERSION, what="fetch", source=None):
    """HBM read (what="fetch") or written (what="write") bytes per launch of `kernel` from the committed rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE pass of this same workload (profiles/r*/pmc_fetch_*.json, pmc_write_*.json; bench.py cannot collect PMCs itself).  A
    profile counts only if it names the same mode (se / pe), read length, genome size and code version; None otherwise -- no claim
    from a stale profile."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{what}_*.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if not isinstance(d, dict) or kernel not in d.get("kernels", {}):
            continue
        if d.get("mode", "se") != mode or d.get("read_len", 150) != read_len or d.get("code_version") != code_version:
            continue
        if abs(d.get("genome_bp", 3.1e9) - total_bp) > 0.02 * total_bp:
            continue
        best = d["kernels"][kernel]["hbm_read_bytes_per_launch" if what == "fetch" else "hbm_write_bytes_per_launch"] * reads_per_launch / d["reads_per_launch"]
        if source is not None:
            source[what] = os.path.relpath(path, ROOT)
    return None if best is None else round(best)


def metric_name(pe, read_len):
    """BASELINE.json's metric for the headline workload (150 bp single-end); the other modes name themselves."""
    if pe:
        return f"reads/s mapped, 2x{read_len} bp PE (-map2) vs hg38-scale index resident in HBM; SAM bit-identical"
    if read_len != 150:
        return f"reads/s mapped, {read_len} bp SE vs hg38-scale index resident in HBM; SAM bit-identical"
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except (OSError, ValueError, KeyError):
        return "reads/s mapped, 150 bp SE vs hg38, at 1/2/4/8 MI355X; SAM bit-identical"


def host_cores():
    """CPUs this process may actually use (affinity and cgroup quota; the GPU boxes show 256 logical CPUs but grant 16)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return cores


def place_index(R, torch, api, device, d_seq, slots, seq_lengths, seq_offsets, labels):
    """The slot table is built by the product's -make_ufi code on the host of rank 0 (UFIndex::MakeIndex is order
    dependent), outside the timed region, and ends up replicated in every rank's HBM: over RCCL (a broadcast of the
    resident arrays from rank 0's GPU -- setup only, nothing on the data path) when every rank has its own GPU, through
    a file in /dev/shm when ranks share a device.  Returns (index, blob_np, seq_np, d_seq, seconds)."""
    t0 = time.time()
    how = "host build + upload"
    blob_np = seq_np = None
    size = int(d_seq.numel())
    cache = os.environ.get("URMAP_BENCH_INDEX_CACHE")  # directory (e.g. /dev/shm/x): reuse the built table between runs
    cpre = os.path.join(cache, f"idx_{CODE_VERSION}_{size}_{slots}") if cache else None
    if R.rank == 0:
        if cpre and os.path.exists(cpre + "_blob.npy"):
            blob_np = np.load(cpre + "_blob.npy", mmap_mode="r")
            seq_np = np.load(cpre + "_seq.npy")
            d_seq = torch.from_numpy(seq_np).to(device)  # the genome the cached table was built from
            how = "cached table"
        else:
            seq_np = d_seq.cpu().numpy()
            if os.environ.get("URMAP_BENCH_HOST_BUILD"):
                blob_np = api.build_slots(seq_np, slots)
            else:  # the product's -make_ufi: counting passes, heads and overflow list on the GPU, ordered inserts on the host
                how = "-make_ufi (GPU passes + ordered inserts on the host) + upload"
                torch.cuda.synchronize()
                blob_np = api.build_slots_gpu(R.device_index, slots, d_seq_ptr=d_seq.data_ptr(), size=size)
            if cpre:
                os.makedirs(cache, exist_ok=True)
                np.save(cpre + "_seq.npy", seq_np)
                np.save(cpre + "_blob.npy", blob_np)
    t_build = time.time() - t0
    t0 = time.time()
    broadcast_s = None
    if R.world == 1:
        index = api.Index.wrap_host(24, 32, slots, blob_np, seq_np, seq_lengths, seq_offsets, labels).upload(R.device_index)
    elif R.backend == "nccl" or os.environ.get("URMAP_BENCH_BROADCAST"):
        # (URMAP_BENCH_BROADCAST: test aid -- take this branch with gloo too, when ranks share a device on a one-GPU box)
        how += f" on rank 0, {'RCCL' if R.backend == 'nccl' else R.backend} broadcast of the resident arrays"
        d_blob = torch.empty(5 * slots + 8, dtype=torch.uint8, device=device)
        d_seqpad = torch.zeros(size + 4096, dtype=torch.uint8, device=device)
        if R.rank == 0:
            with warnings.catch_warnings():  # a cached table is mapped read-only; it is only copied to the device
                warnings.simplefilter("ignore", UserWarning)
                step = 1 << 30
                for lo in range(0, 5 * slots, step):
                    hi = min(5 * slots, lo + step)
                    d_blob[lo:hi] = torch.from_numpy(np.asarray(blob_np[lo:hi])).to(device)
            d_blob[5 * slots:] = 0
            d_seqpad[:size] = d_seq
        t_b = time.time()
        R.broadcast_bytes(torch, d_blob)
        R.broadcast_bytes(torch, d_seqpad)
        torch.cuda.synchronize()
        broadcast_s = time.time() - t_b
        d_seq = d_seqpad[:size]
        index = api.Index.wrap_device(R.device_index, 24, 32, slots, d_blob.data_ptr(), d_seqpad.data_ptr(), size,
                                      seq_lengths, seq_offsets, labels, keep=(d_blob, d_seqpad))
    else:
        how += " on rank 0, shared through a file"
        shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        shm = os.path.join(shm_dir, f"urmap_bench_{os.environ.get('MASTER_PORT', 'solo')}_{size}")
        if R.rank == 0:
            np.save(shm + "_blob.npy", blob_np)
            np.save(shm + "_seq.npy", seq_np)
        R.barrier()
        if R.rank != 0:
            blob_r = np.load(shm + "_blob.npy", mmap_mode="r")
            seq_r = np.load(shm + "_seq.npy", mmap_mode="r")
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)
                d_seq = torch.from_numpy(np.ascontiguousarray(seq_r)).to(device)
            index = api.Index.wrap_host(24, 32, slots, blob_r, seq_r, seq_lengths, seq_offsets, labels).upload(R.device_index)
        else:
            index = api.Index.wrap_host(24, 32, slots, blob_np, seq_np, seq_lengths, seq_offsets, labels).upload(R.device_index)
        R.barrier()
        if R.rank == 0:
            for suf in ("_blob.npy", "_seq.npy"):
                try:
                    os.remove(shm + suf)
                except OSError:
                    pass
    info = {"make_ufi": round(t_build, 1), "upload": round(time.time() - t0, 1), "how": how,
            # what every rank holds in HBM for the index: slot table + sequence + its packed copy (4 bit planes per 32 bases)
            # + every chain head's row laid out beside the table (chain_rows.hip: 4 B per slot + 4 B per chained position)
            "index_bytes_per_rank": int(5 * slots + 8 + size + 4096 + 16 * ((size + 31) // 32 + 1)) + int(index.chain_row_bytes()),
            "chain_row_bytes": int(index.chain_row_bytes())}
    if broadcast_s is not None:
        info["broadcast_s"] = round(broadcast_s, 2)
        info["broadcast_pieces"] = int((5 * slots + 8 + (1 << 30) - 1) >> 30) + int((size + 4096 + (1 << 30) - 1) >> 30)
    return index, blob_np, seq_np, d_seq, info


class Workload:
    """One synthetic read set resident in HBM + the output arrays of a step.  A batch is mapped the way the command line
    maps it: split over `streams` mapping contexts of the device (urmap -streams K, default 2), each with its own HIP
    stream, so that one context's probe / DP / finalize launches overlap the other's search kernel."""

    def __init__(self, torch, api, device, d_seq, seq_lengths, seq_offsets, pe, L, sub, indel, nb, n_batches, seed, streams=1, contexts=1):
        self.pe, self.L, self.nb, self.api, self.torch = pe, L, nb, api, torch
        self.contexts = max(1, contexts)
        if self.contexts > 1:
            streams = 1  # whole batches alternate over the contexts (round 6); a batch is not split
        self.batches = []
        for b in range(n_batches):
            if pe:
                self.batches.append(make_pairs_torch(torch, seed + b, d_seq, seq_lengths, seq_offsets, nb // 2, L, sub, 1.5 * sub, device))
            else:
                self.batches.append(make_reads_torch(torch, seed + b, d_seq, seq_lengths, seq_offsets, nb, L, sub, indel, device))
        # contiguous parts of the batch, one per context (pairs stay together)
        unit = 2 if pe else 1
        cuts = [((nb // unit) * k // streams) * unit for k in range(streams + 1)]
        self.parts = []
        for k in range(streams):
            n = cuts[k + 1] - cuts[k]
            self.parts.append({"lo": cuts[k], "n": n,
                               "d_offs": (torch.arange(n + 1, device=device, dtype=torch.int64) * L).contiguous(),
                               "d_results": torch.zeros(n * api.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=device),
                               "d_pathops": torch.zeros(n * api.MAX_PATH_OPS, dtype=torch.int16, device=device),
                               "d_used": torch.zeros(1, dtype=torch.int32, device=device)})
        # round 6: `contexts` > 1 -- whole batches ALTERNATE over that many mapping contexts of the device, each with its own HIP stream and its own
        # output arrays, as the lanes of urmapx_map_files (urmap -streams K, default 2) run a device: context B's search kernel fills the CUs that
        # context A's tail and phase-6 launches leave idle.  Every step is still one pass of the hot path over one whole batch.
        self.outs = [self.parts[0]]
        for _ in range(1, self.contexts):
            p0 = self.parts[0]
            self.outs.append({"lo": 0, "n": p0["n"], "d_offs": p0["d_offs"], "d_results": torch.zeros_like(p0["d_results"]),
                              "d_pathops": torch.zeros_like(p0["d_pathops"]), "d_used": torch.zeros_like(p0["d_used"])})
        self.last = None

    def step_on(self, m, c, b):
        """batch b on context c (alternating mode): launches only, no wait"""
        self.last = self.batches[b % len(self.batches)]
        self.parts = [self.outs[c]]
        p = self.outs[c]
        f = m.map_pe_device if self.pe else m.map_se_device
        f(self.last.data_ptr(), p["d_offs"].data_ptr(), p["n"] // 2 if self.pe else p["n"], p["n"] * self.L, self.L,
          p["d_results"].data_ptr(), p["d_pathops"].data_ptr(), p["d_used"].data_ptr())

    def step(self, mappers, b):
        self.last = self.batches[b % len(self.batches)]
        for m, p in zip(mappers, self.parts):
            f = m.map_pe_device if self.pe else m.map_se_device
            f(self.last.data_ptr() + p["lo"] * self.L, p["d_offs"].data_ptr(), p["n"] // 2 if self.pe else p["n"], p["n"] * self.L, self.L,
              p["d_results"].data_ptr(), p["d_pathops"].data_ptr(), p["d_used"].data_ptr())

    def timed(self, mappers, steps, warmup, barrier=None):
        """W untimed steps, then K timed ones -> (seconds, mean ms per launch [probe, search]); stage_ms / dp_stats too"""
        alternating = self.contexts > 1 and len(list(mappers)) >= self.contexts
        mappers = list(mappers)[: self.contexts] if alternating else list(mappers)[: len(self.parts)]
        self.torch.cuda.synchronize()  # the batches were written on torch's stream; the contexts run on streams of their own
        # (alternating: every context takes at least one untimed step when any warm-up is asked for -- its first call allocates its device arrays)
        self.warmup_run = max(warmup, self.contexts) if alternating and warmup > 0 else warmup
        for w in range(self.warmup_run):
            if alternating:
                self.step_on(mappers[w % self.contexts], w % self.contexts, w)
            else:
                self.step(mappers, w)
        for m in mappers:
            m.sync()
        if barrier:
            barrier()
        kms = np.zeros(2)
        self.stage_ms = np.zeros(7)
        self.round_ms = None
        self.p3_ms, self.p3_stats = np.zeros(3), np.zeros(2)
        share = 1.0 if alternating else 1.0 / len(mappers)

        def collect(m):  # the launches of the step this context has just finished, by the events on its own stream
            nonlocal kms
            kms += np.array(m.last_kernel_ms()) * share
            if not self.pe:
                self.stage_ms += np.array(m.stage_ms()) * share
                rm = np.array(m.round_ms()) * share / max(1, steps)
                self.round_ms = rm if self.round_ms is None else self.round_ms + rm
                p3m, p3s = m.phase3()
                self.p3_ms += np.array(p3m) * share / max(1, steps)
                self.p3_stats += np.array(p3s, dtype=np.float64) / max(1, steps)
        t0 = time.perf_counter()
        if alternating:
            C = self.contexts
            for k in range(steps):
                c = k % C
                if k >= C:  # this context's step k - C must be done before its arrays are reused; the other contexts run on
                    mappers[c].sync()
                    collect(mappers[c])
                self.step_on(mappers[c], c, warmup + k)
            for k in range(max(0, steps - C), steps):  # the last step of every context, oldest first: the one that ran step K - 1 is waited for last
                mappers[k % C].sync()
                collect(mappers[k % C])
            last_m = [mappers[(steps - 1) % C]]
        else:
            for k in range(steps):
                self.step(mappers, warmup + k)
                for m in mappers:
                    m.sync()
                for m in mappers:
                    collect(m)
            last_m = mappers
        self.own_s = time.perf_counter() - t0  # this rank's K steps, before it waits for the others
        if barrier:
            barrier()
        self.stage_ms /= max(1, steps)
        self.dp_stats = None if self.pe else [int(x) for x in np.sum([m.dp_stats() for m in last_m], axis=0)]
        return time.perf_counter() - t0, kms / max(1, steps)

    def results(self):
        """results of the last step, parts joined, path offsets made to index one joined path arena"""
        api = self.api
        res, ops, base = [], [], 0
        for p in self.parts:
            g = np.frombuffer(p["d_results"].cpu().numpy().tobytes(), dtype=api.RESULT_DTYPE).copy()
            used = int(p["d_used"].cpu().item())
            o = p["d_pathops"][:used].cpu().numpy().view(np.uint16)
            g["path_off"] += np.uint32(base)
            base += used
            res.append(g)
            ops.append(o)
        return np.concatenate(res), np.concatenate(ops) if ops else np.zeros(0, np.uint16)

    def check(self, oi, sample_n, threads):
        """The last step's results against the oracle on its first sample_n reads: every field SAM is made of (position,
        strand, scores, MAPQ, the alignment path) -> (parity dict, oracle counters per read, oracle seconds)."""
        api, L = self.api, self.L
        sample_n = int(min(self.nb, sample_n)) & ~1
        hb = self.last[: sample_n * L].cpu().numpy()
        ho = (np.arange(sample_n + 1, dtype=np.uint64) * L)
        t1 = time.perf_counter()
        ores, opaths, cnt = oi.map_pe(hb, ho, threads=threads) if self.pe else oi.map_se(hb, ho, threads=threads)
        t_cpu = time.perf_counter() - t1
        g, gops = self.results()
        g = g[:sample_n]
        diffs = {"status_nonzero": int((g["status"] != 0).sum())}
        ok = diffs["status_nonzero"] == 0
        for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq"):
            nd = int((g[name].astype(np.int64) != ores[name].astype(np.int64)).sum())
            diffs[name] = nd
            ok = ok and nd == 0
        mapped = ores["dbpos"] != 0xFFFFFFFF
        diffs["plus"] = int((g["plus"][mapped] != ores["plus"][mapped]).sum())
        # paths: ungapped hits carry no path on either side; gapped ones are compared run for run
        gapped = np.nonzero(mapped & ((g["path_nops"] > 0) | (ores["path_len"] > 0)))[0]
        bad_paths = 0
        for i in gapped:
            o = int(g["path_off"][i])
            if api.decode_path(gops[o:o + int(g["path_nops"][i])]) != opaths[i]:
                bad_paths += 1
        diffs["path"] = bad_paths
        ok = ok and diffs["plus"] == 0 and bad_paths == 0
        parity = {"reads_checked": int(sample_n), "bit_identical_to_oracle": bool(ok),
                  "fields": "dbpos, seq_index, coord, plus, score, second, mapq, path (CIGAR)",
                  "gapped_paths_checked": int(len(gapped)), "mapped_frac": round(float(mapped.mean()), 4)}
        if not self.pe:  # the phase of Search_Lo a read left in (search1m6.cpp:35-277): what share of the batch each phase sees
            h = np.bincount(ores["exit_phase"].astype(np.int64), minlength=7)[:7]
            parity["exit_phase_frac"] = [round(float(x) / max(1, sample_n), 4) for x in h]
            parity["exit_phase_equal"] = bool((g["exit_phase"] == ores["exit_phase"]).all())
        if not ok:
            parity["mismatches"] = diffs
            parity["status_values"] = [int(x) for x in np.unique(g["status"])]
        return parity, {k: v / max(1, cnt["n_reads"]) for k, v in cnt.items()}, t_cpu


def run_timed(wl, mappers, steps, warmup, barrier=None, alone_steps=3):
    """wl.timed over the contexts the run uses, and -- when batches alternate over several -- a few steps of ONE context with the device to itself afterwards
    (outside the timed region): what each launch takes when nothing shares the device with it.  -> (seconds, kms, alone | None)"""
    dt, kms = wl.timed(mappers, steps, warmup, barrier)
    alone = None
    if wl.contexts > 1 and alone_steps > 0 and len(mappers) >= wl.contexts:
        keep = (wl.stage_ms.copy(), wl.round_ms, wl.p3_ms.copy(), wl.p3_stats.copy(), wl.dp_stats, wl.own_s)
        C, wl.contexts = wl.contexts, 1
        adt, akms = wl.timed([mappers[0]], alone_steps, 1)
        alone = {"ms_per_step": 1e3 * adt / alone_steps, "reads_per_s": alone_steps * wl.nb / adt, "kms": akms, "stage_ms": wl.stage_ms.copy(), "p3_ms": wl.p3_ms.copy(),
                 "steps": alone_steps}
        wl.contexts = C
        wl.stage_ms, wl.round_ms, wl.p3_ms, wl.p3_stats, wl.dp_stats, wl.own_s = keep
    return dt, kms, alone


def alone_block(alone, contexts):
    return None if alone is None else {
        "value": round(alone["reads_per_s"], 1), "unit": "reads/s", "ms_per_step": round(alone["ms_per_step"], 3), "steps": alone["steps"],
        "what": f"ONE mapping context running its steps back to back with the device to itself (the timed region of rounds 1-5), measured after the timed region of this run, "
                f"in which whole batches alternate over {contexts} contexts: the kernels' `alone_ms` come from these steps"}


def write_fastq_fixed(path, reads_u8, n, L, first=0):
    """n reads of length L (uint8 array [n*L]) as FASTQ with fixed-width labels r00000000.. (counted from `first`) and quality 'I'; `path` may be an open file."""
    lab = 2 + 8 + 1
    rec = lab + L + 3 + L + 1
    a = np.empty((n, rec), dtype=np.uint8)
    a[:, 0] = ord("@"); a[:, 1] = ord("r")
    idx = np.arange(first, first + n, dtype=np.int64)
    for d in range(8):
        a[:, 2 + d] = ((idx // 10 ** (7 - d)) % 10 + ord("0")).astype(np.uint8)
    a[:, 10] = ord("\n")
    a[:, lab:lab + L] = reads_u8.reshape(n, L)
    a[:, lab + L] = ord("\n"); a[:, lab + L + 1] = ord("+"); a[:, lab + L + 2] = ord("\n")
    a[:, lab + L + 3:lab + 2 * L + 3] = ord("I")
    a[:, rec - 1] = ord("\n")
    a.tofile(path)
    return rec * n


def run_reference(ref_bin, oi, d, fq, fq_head, n_reads, n_head, cores, product_sam):
    """The reference binary itself (oracle/_ref/urmap, compiled from /root/reference by oracle/Makefile; it travels with
    the snapshot as a test tool) on this host: the index is written as a .ufi file in /dev/shm, `urmap -map` runs on the
    whole FASTQ file and on its head, the difference of the two wall times is mapping time without the index load.  Its
    SAM for the head must equal the product's records (as a set: the reference writes in completion order)."""
    import subprocess
    ufi = os.path.join(d, "idx.ufi")
    t0 = time.time()
    oi.save(ufi)
    t_save = time.time() - t0

    def run(fastq, out):
        t = time.time()
        r = subprocess.run([ref_bin, "-map", fastq, "-ufi", ufi, "-samout", out, "-threads", str(cores)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            raise RuntimeError("reference urmap failed: " + r.stderr.decode()[-500:])
        return time.time() - t
    sam_head, sam_all = os.path.join(d, "ref_head.sam"), os.path.join(d, "ref_all.sam")
    t_head = run(fq_head, sam_head)
    t_all = run(fq, sam_all)
    os.remove(sam_all)
    os.remove(ufi)
    want = sorted(l for l in open(sam_head, "rb").read().split(b"\n") if l and not l.startswith(b"@"))
    got = []
    with open(product_sam, "rb") as f:
        for line in f:
            if line.startswith(b"@"):
                continue
            got.append(line.rstrip(b"\n"))
            if len(got) == len(want):
                break
    rate = (n_reads - n_head) / max(1e-9, t_all - t_head)
    return {"reads_per_s": round(rate, 1), "threads": cores, "wall_all_reads_s": round(t_all, 1), "wall_head_s": round(t_head, 1),
            "how": f"wall({n_reads} reads) - wall({n_head} reads) = {t_all:.1f} - {t_head:.1f} s (each run loads the index from /dev/shm); "
                   f".ufi written in {t_save:.1f} s",
            "sam_records_identical_to_product": bool(sorted(got) == want), "sam_records_checked": len(want)}


def sam_head_records(path, n):
    """the first n records (lines that are not header lines) of a SAM file"""
    got = []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b"@"):
                continue
            got.append(line.rstrip(b"\n"))
            if len(got) == n:
                break
    return got


def sam_tail_records(path, n, per_record=700):
    """the last n records of a SAM file, read from its end (the file may be tens of GB)"""
    size = os.path.getsize(path)
    want = n * per_record
    while True:
        with open(path, "rb") as f:
            f.seek(max(0, size - want))
            lines = f.read().split(b"\n")[:-1]
        if len(lines) > n or want >= size:
            return [l for l in lines if not l.startswith(b"@")][-n:]
        want *= 2


def write_ceiling_gbs(d, total=1 << 30, piece=64 << 20):
    """What one thread's pwrite reaches on the medium the SAM file is written to (the same directory): GB/s."""
    buf = bytes(piece)
    path = os.path.join(d, "write_probe.bin")
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        t0 = time.perf_counter()
        off = 0
        while off < total:
            off += os.pwrite(fd, buf, off)
        dt = time.perf_counter() - t0
    finally:
        os.close(fd)
        os.remove(path)
    return total / dt / 1e9


def bgzf_bytes(data, block=60000, level=1):
    """BGZF container (bgzip / htslib): gzip members of <= 64 KB with their size in a 'BC' extra field."""
    import struct
    import zlib
    out = bytearray()
    for i in list(range(0, len(data), block)) + [len(data)]:
        raw = data[i:i + block] if i < len(data) else b""
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 18 + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    return bytes(out)


def files_equal_concat(one, parts, block=64 << 20):
    """`cat parts` == the file `one`, byte for byte"""
    with open(one, "rb") as f:
        for p in parts:
            with open(p, "rb") as g:
                while True:
                    b = g.read(block)
                    if not b:
                        break
                    if f.read(len(b)) != b:
                        return False
        return f.read(1) == b""


def e2e_bound(rep):
    """Which stage of urmapx_map_files' pipeline the run waited for: the stage whose busy time fills the wall time."""
    wall = max(rep["seconds"], 1e-9)
    lanes = max(1, rep["lanes"])
    shares = {"output file write (one pwrite stream into " + rep["medium"].decode() + ")": rep["write_s"] / wall,
              "device lanes (copies + kernels)": rep["gpu_s"] / lanes / wall,
              "input read / inflate": rep["parse_s"] / wall,
              "host SAM formatting": rep["format_s"] / wall}
    name = max(shares, key=shares.get)
    return name, {k: round(v, 3) for k, v in shares.items()}


E2E_STREAMS = int(os.environ.get("URMAP_BENCH_E2E_STREAMS", 2))  # mapping contexts (lanes) of the file-to-file runs
E2E_BATCH = int(os.environ.get("URMAP_BENCH_E2E_BATCH", 0))      # reads per chunk (0: the library chooses: 262 144 .. 1 M by the size of the file)


def _cgroup_cpu():
    """cpu.stat of this process's cgroup (v2): periods in which the CPU quota ran out, and for how long its threads were stopped"""
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):  # v2, v1
        try:
            for line in open(path):
                k, v = line.split()
                if k in ("nr_periods", "nr_throttled", "throttled_usec", "usage_usec"):
                    out[k] = int(v)
                elif k == "throttled_time":  # v1: nanoseconds
                    out["throttled_usec"] = int(v) // 1000
            break
        except (OSError, ValueError):
            continue
    return out


def watched(fn):
    """fn() with what the host side did meanwhile: CPU seconds of this process, the cgroup's throttled periods / time"""
    c0, t0, w0 = _cgroup_cpu(), os.times(), time.time()
    rep = fn()
    c1, t1, w1 = _cgroup_cpu(), os.times(), time.time()
    rep["host"] = {"process_cpu_s": round((t1.user - t0.user) + (t1.system - t0.system), 3), "call_wall_s": round(w1 - w0, 3),
                   "cgroup_throttled_periods": c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0),
                   "cgroup_throttled_ms": round((c1.get("throttled_usec", 0) - c0.get("throttled_usec", 0)) / 1e3, 1),
                   "cgroup_cpu_s": round((c1.get("usage_usec", 0) - c0.get("usage_usec", 0)) / 1e6, 3)}
    return rep


def lane_view(rep):
    """where the lanes' time went in one urmapx_map_files run (seconds summed over lanes)"""
    return {"stream_time_s": {k[4:-2]: round(rep[k], 3) for k in ("dev_h2d_s", "dev_parse_s", "dev_map_s", "dev_format_s", "dev_d2h_s")},
            "map_search_s": round(rep["dev_map_search_s"], 3), "map_dp_finalize_s": round(rep["dev_map_dp_s"], 3),
            "map_enqueue_host_s": round(rep["map_enqueue_s"], 3), "lane_busy_s": round(rep["gpu_s"], 3),
            "alloc": {"device_s": round(rep["alloc_dev_s"], 3), "device_calls": rep["alloc_dev_calls"], "pinned_s": round(rep["alloc_pinned_s"], 3), "pinned_calls": rep["alloc_pinned_calls"]},
            "host": rep.get("host")}


def run_cli(oi, d, fq, n_reads, want, ref):
    """What a user of the command line gets, index load included: `urmap -map reads.fq -ufi index.ufi -samout out.sam` as a process of its
    own (the .ufi in /dev/shm, as for the reference binary's run): wall time of the process, its own "Seconds to load index" and
    "Seconds in mapper", with the index streamed from the file to the device (urmapx_index_open_device, what the command line does) and,
    for comparison, read into host arrays first (URMAPX_HOST_INDEX=1: the loader of rounds 1-4).  The first records of its SAM must be the oracle's."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "urmap_amd", "urmap")
    ufi, sam = os.path.join(d, "cli.ufi"), os.path.join(d, "cli.sam")
    t0 = time.time()
    oi.save(ufi)
    out = {"ufi_GB": round(os.path.getsize(ufi) / 1e9, 2), "ufi_written_in_s": round(time.time() - t0, 1), "reads": n_reads}
    # (the first process to read the freshly written .ufi out of /dev/shm reads it at a third of the rate of the ones after it: it is kept apart)
    variants = [("index_streamed_first_read_of_the_file", {}), ("index_through_host_arrays", {"URMAPX_HOST_INDEX": "1"}), ("index_streamed_to_the_device", {})]
    if os.environ.get("URMAP_BENCH_CLI_THREADS"):  # measurement: the loader's reader threads
        variants = [(f"streamed_with_{t}_reader_threads", {"URMAPX_LOAD_THREADS": t}) for t in os.environ["URMAP_BENCH_CLI_THREADS"].split(",")]
    settle = float(os.environ.get("URMAP_BENCH_CLI_SETTLE_S", 8))
    out["settle_s_before_each_run"] = settle
    for name, extra in variants:
        env = dict(os.environ, OMP_WAIT_POLICY="passive", URMAPX_VERBOSE="1", **extra)
        # The device clears memory a process has freed (122 GB here: this process's replica, then the run before), and a process that
        # starts within seconds of that waits for it in its first allocations (measured: the same command 4-6 s instead of 2.5-4 s).
        # A user's run starts on an idle device: each run here waits a few seconds first.
        time.sleep(settle)
        t = time.time()
        r = subprocess.run([exe, "-map", fq, "-ufi", ufi, "-samout", sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        wall = time.time() - t
        text = (r.stdout + r.stderr).decode("latin-1")
        if r.returncode != 0:
            out[name] = {"error": text[-300:]}
            continue
        num = lambda what: (lambda m: float(m.group(1)) if m else None)(re.search(r"([0-9.]+)\s+" + what, text))
        got = []
        with open(sam, "rb") as f:
            for line in f:
                if line.startswith(b"@"):
                    continue
                got.append(line.rstrip(b"\n"))
                if len(got) == len(want):
                    break
        m = re.search(r"index streamed to device \d+: ([0-9.]+) GB in ([0-9.]+) s \((\d+) reader threads\), resident layouts built in ([0-9.]+) s", text)
        if m:
            out["load_parts_s"] = {"file_to_device": float(m.group(2)), "resident_layouts": float(m.group(4)), "reader_threads": int(m.group(3))}  # (of the last streamed run)
            if os.environ.get("URMAP_BENCH_CLI_THREADS"):
                out.setdefault("file_to_device_s_by_threads", {})[m.group(3)] = float(m.group(2))
        out[name] = {"wall_s": round(wall, 2), "seconds_to_load_index": num("Seconds to load index"), "seconds_in_mapper": num("Seconds in mapper"),
                     "reads_per_s_of_wall": round(n_reads / wall, 1), "sam_records_identical_to_oracle": bool(got == want)}
        os.remove(sam)
    os.remove(ufi)
    if ref and ref.get("wall_all_reads_s"):
        out["reference_binary_wall_s"] = ref["wall_all_reads_s"]
        w = out.get("index_streamed_to_the_device", {}).get("wall_s")
        if w:
            out["reference_wall_over_product_wall"] = round(ref["wall_all_reads_s"] / w, 1)
    return out


def run_e2e(torch, api, oi, index, device, d_seq, seq_lengths, seq_offsets, L, sub, indel, n_reads, cores, ref_bin=None, gpus=1, release_device=None):
    """FASTQ file -> SAM file through urmapx_map_files (the command line's cmd_map) on the resident index: what a user of
    `urmap -map` gets, index load excluded as the reference reports it.  Files live in /dev/shm (memory), so this is the
    read + PCIe + parse + map + format + write pipeline, not a disk benchmark.  Chunks of the FASTQ file go to the device
    as bytes and come back as the bytes of their SAM records (text_gpu.hip); the host reads and writes.  The first 100 k records are compared with the
    SAM the oracle writes for the same reads."""
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="urmap_e2e_", dir=base)
    try:
        fq, sam = os.path.join(d, "r.fq"), os.path.join(d, "out.sam")
        # the reads a slab at a time (the generator's index arrays are 8 bytes per base: 100 M reads -- the N = 8 leg -- at once would not fit), labels counted through
        slab = 2_000_000
        with open(fq, "wb") as f:
            for lo in range(0, n_reads, slab):
                k = min(slab, n_reads - lo)
                part = make_reads_torch(torch, 777 + lo // slab, d_seq, seq_lengths, seq_offsets, k, L, sub, indel, device).cpu().numpy()
                write_fastq_fixed(f, part, k, L, first=lo)
                del part
        fq_bytes = os.path.getsize(fq)
        n_chk = min(n_reads // 4, 400_000)
        fq_head, sam_o = os.path.join(d, "head.fq"), os.path.join(d, "oracle.sam")
        with open(fq, "rb") as f, open(fq_head, "wb") as g:
            g.write(f.read(n_chk * (fq_bytes // n_reads)))
        # the LAST n_chk reads too (chunk numbers, byte offsets and path-arena offsets are largest at the end of the file);
        # their labels are the file's (write_fastq_fixed counts from 0: the tail file is cut out of the big one below)
        fq_tail, sam_ot = os.path.join(d, "tail.fq"), os.path.join(d, "oracle_tail.sam")
        rec_bytes = fq_bytes // n_reads
        with open(fq, "rb") as f, open(fq_tail, "wb") as g:
            f.seek((n_reads - n_chk) * rec_bytes)
            g.write(f.read())
        runs = []
        for _ in range(2):  # the second run has its buffers and the page cache warm; both are reported
            rep = watched(lambda: api.map_files(index, fq, samout=sam, first_gpu=device.index, gpus=gpus, streams=E2E_STREAMS, batch=E2E_BATCH, cmdline="bench.py e2e"))
            runs.append(rep)
        rep = runs[-1]
        oi.map_file_se(fq_head, sam_o, threads=cores)
        ref = None
        if gpus == 1 and ref_bin and os.path.exists(ref_bin) and not os.environ.get("URMAP_BENCH_NO_REFERENCE"):
            try:
                ref = run_reference(ref_bin, oi, d, fq, fq_head, n_reads, n_chk, cores, sam)
            except Exception as e:  # the baseline is reported when it can be had; the measurement does not depend on it
                ref = None
                print(f"bench.py: reference binary not timed: {e}", file=sys.stderr)
        want = [l for l in open(sam_o, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        same = sam_head_records(sam, len(want)) == want
        oi.map_file_se(fq_tail, sam_ot, threads=cores)
        want_tail = [l for l in open(sam_ot, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        same_tail = len(want_tail) == n_chk and sam_tail_records(sam, n_chk) == want_tail
        os.remove(fq_tail); os.remove(sam_ot)
        bound, shares = e2e_bound(rep)
        # the same run with the SAM text dropped after it has reached the host: what the device lanes sustain when the output
        # medium is out of the way, and where a lane's time goes (events on the lanes' streams)
        null_reps = [watched(lambda: api.map_files(index, fq, samout=sam + ".null", first_gpu=device.index, gpus=gpus, streams=E2E_STREAMS, batch=E2E_BATCH,
                                                   cmdline="bench.py e2e", discard_sam=True)) for _ in range(3)]
        null_rep = null_reps[-1]  # (three calls, the last one reported: the first two still warm buffers and lanes up -- scripts/r6_lanes.py shows the same run after run)
        lanes = max(1, null_rep["lanes"])
        null_sink = {"value": round(null_rep["reads"] / null_rep["seconds"], 1), "unit": "reads/s", "seconds": round(null_rep["seconds"], 3),
                     "all_runs_reads_per_s": [round(r["reads"] / r["seconds"], 1) for r in null_reps],
                     "lanes": lanes, "lane_busy_s_summed": round(null_rep["gpu_s"], 3),
                     "stream_time_s_summed_over_lanes": {k[4:-2]: round(null_rep[k], 3) for k in ("dev_h2d_s", "dev_parse_s", "dev_map_s", "dev_format_s", "dev_d2h_s")},
                     "lanes_view": lane_view(null_rep),
                     "note": "urmapx_map_files with discard_sam: FASTQ bytes to the device, SAM bytes back to the host, nothing written; "
                             "stream times from HIP events per chunk (copy in, line ends + record checks + base copy, mapping kernels, SAM lengths + text, copy out), "
                             "summed over the chunks of all lanes: divide by `lanes` for the wall share"}
        # one SAM file per pipeline (urmap -samshards N): N readers, lanes and writers side by side; `cat` of the shards must be the one file
        n_shards = gpus if gpus > 1 else 2
        for p in [sam + f".sh.{k}" for k in range(n_shards)]:
            if os.path.exists(p):
                os.remove(p)
        sh_rep = [watched(lambda: api.map_files(index, fq, samout=sam + ".sh", first_gpu=device.index, gpus=gpus, streams=E2E_STREAMS, batch=E2E_BATCH,
                                                cmdline="bench.py e2e", sam_shards=n_shards)) for _ in range(2)][-1]
        sharded = {"value": round(sh_rep["reads"] / sh_rep["seconds"], 1), "unit": "reads/s", "seconds": round(sh_rep["seconds"], 3), "shards": n_shards,
                   "vs_one_file": round((sh_rep["reads"] / sh_rep["seconds"]) / (rep["reads"] / rep["seconds"]), 3),
                   "cat_of_shards_equals_the_one_file": files_equal_concat(sam, [sam + f".sh.{k}" for k in range(n_shards)]),
                   "lanes_view": lane_view(sh_rep),
                   "note": f"urmap -samout out.sam -samshards {n_shards}: shard s = the s-th part of the input (cut at a record), its own reader, "
                           f"lanes and writer{'' if gpus > 1 else ' (here: two pipelines on the one GPU)'}; header in shard 0"}
        for k in range(n_shards):
            os.remove(sam + f".sh.{k}")
        ceiling = write_ceiling_gbs(d)
        sam_gb = os.path.getsize(sam) / 1e9
        out = {"value": round(rep["reads"] / rep["seconds"], 1), "unit": "reads/s", "reads": int(rep["reads"]), "gpus": gpus,
               "bound": bound, "stage_share_of_wall": shares,
               "output_medium": {"kind": rep["medium"].decode(), "one_thread_pwrite_GBs": round(ceiling, 2), "writer_threads": int(rep["write_threads"]),
                                 "reads_per_s_at_that_ceiling": round(rep["reads"] / (sam_gb / ceiling), 1),
                                 "note": "measured in this run: 1 GiB written with pwrite by one thread into the directory of the SAM file"},
               "seconds": round(rep["seconds"], 3), "first_run_seconds": round(runs[0]["seconds"], 3),
               "what": f"urmapx_map_files (= urmap -map): {fq_bytes / 1e9:.2f} GB FASTQ file -> {os.path.getsize(sam) / 1e9:.2f} GB SAM file, both in /dev/shm; "
                       f"index resident, {rep['host_threads']} host threads, {rep['lanes']} mapping contexts on one GPU; FASTQ parsing and SAM formatting on the device "
                       f"(format_s 0 = no host formatting; write_s = one thread's pwrite into tmpfs, the stage that bounds the run)",
               "stage_busy_s": {k: round(rep[k], 3) for k in ("parse_s", "gpu_s", "format_s", "write_s")},
               "mapped_q10_frac": round(rep["mapped_q"] / max(1, rep["reads"]), 4),
               "lanes_view": lane_view(rep), "first_run_lanes_view": lane_view(runs[0]),
               "placement": rep["placement"].decode(),  # NUMA node of each device's PCI function = where its lane threads ran (@any: not pinned)
               "sam_records_identical_to_oracle": bool(same and same_tail), "sam_records_checked": len(want) + len(want_tail),
               "sam_slices_checked": {"head": [0, len(want)], "tail": [n_reads - len(want_tail), n_reads], "head_identical": bool(same), "tail_identical": bool(same_tail)},
               "null_sink": null_sink, "sharded": sharded}
        if ref:
            out["reference_binary"] = ref
        if gpus == 1 and not os.environ.get("URMAP_BENCH_NO_E2E_GZ"):
            out["gz"] = run_e2e_gz(api, index, device, d, fq, n_reads, L, want)
        if gpus == 1 and not os.environ.get("URMAP_BENCH_NO_E2E_PAIRS"):
            out["pairs"] = run_e2e_pairs(torch, api, oi, index, device, d_seq, seq_lengths, seq_offsets, L,
                                         int(os.environ.get("URMAP_BENCH_E2E_PAIRS", n_reads)), cores, d)  # BASELINE config 3: 10 M PAIRS
        if gpus == 1 and release_device is not None and not os.environ.get("URMAP_BENCH_NO_CLI"):
            release_device()  # the command line's process loads an index of its own: this process's replica and contexts leave the device first
            try:
                out["cli"] = run_cli(oi, d, fq, n_reads, want, ref)
            except Exception as e:
                out["cli"] = {"error": str(e)[:300]}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_e2e_gz(api, index, device, d, fq, n_reads, L, want):
    """The same FASTQ compressed: `urmap -map reads.fq.gz`.  The reader thread inflates (one gzip member: one zlib stream;
    a BGZF file: blocks in parallel on the host threads), cuts chunks out of the inflated text and the device parses and
    formats as for a plain file.  The first records of the SAM must be the plain run's."""
    import subprocess
    n_gz = min(n_reads, int(os.environ.get("URMAP_BENCH_E2E_GZ_READS", 2_000_000)))
    rec = (2 + 8 + 1) + L + 3 + L + 1
    with open(fq, "rb") as f:
        data = f.read(rec * n_gz)
    out = {}
    for kind in ("gzip", "bgzf"):
        gz = os.path.join(d, kind + ".fq.gz")
        t0 = time.time()
        if kind == "gzip":
            with open(gz, "wb") as f:
                subprocess.run(["gzip", "-1", "-c"], input=data, stdout=f, check=True)
        else:
            with open(gz, "wb") as f:
                f.write(bgzf_bytes(data))
        t_make = time.time() - t0
        sam = os.path.join(d, kind + ".sam")
        reps = [watched(lambda: api.map_files(index, gz, samout=sam, first_gpu=device.index, gpus=1, streams=E2E_STREAMS, batch=E2E_BATCH, cmdline="bench.py e2e gz")) for _ in range(2)]
        rep = reps[-1]
        got = []
        with open(sam, "rb") as f:
            for line in f:
                if line.startswith(b"@"):
                    continue
                got.append(line.rstrip(b"\n"))
                if len(got) == min(len(want), n_gz):
                    break
        bound, shares = e2e_bound(rep)
        out[kind] = {"value": round(rep["reads"] / rep["seconds"], 1), "unit": "reads/s", "reads": int(rep["reads"]), "seconds": round(rep["seconds"], 3),
                     "compressed_GB": round(os.path.getsize(gz) / 1e9, 3), "fastq_GB": round(len(data) / 1e9, 3),
                     "inflate_GBs": round(rep["input_bytes"] / max(rep["parse_s"], 1e-9) / 1e9, 2),
                     "text_on_device": bool(rep["text_on_device"]), "format_s": round(rep["format_s"], 3), "bound": bound, "stage_share_of_wall": shares,
                     "sam_records_identical_to_plain_run": bool(got == want[: len(got)]), "made_in_s": round(t_make, 1), "lanes_view": lane_view(rep)}
        os.remove(gz)
        os.remove(sam)
    return out


def run_e2e_pairs(torch, api, oi, index, device, d_seq, seq_lengths, seq_offsets, L, npairs, cores, d):
    """`urmap -map2`: two mate files -> SAM through urmapx_map_files (cmd_map2) on the resident index, both mate files'
    chunks parsed and the pair records written on the device.  BASELINE config 3: 10 M pairs.  The head AND the tail of the SAM are
    compared with the oracle's."""
    fq1, fq2, sam = os.path.join(d, "m1.fq"), os.path.join(d, "m2.fq"), os.path.join(d, "pairs.sam")
    n_chk = min(npairs // 4, 100_000)
    h1, h2, sam_o = os.path.join(d, "h1.fq"), os.path.join(d, "h2.fq"), os.path.join(d, "pairs_oracle.sam")
    t1, t2 = os.path.join(d, "t1.fq"), os.path.join(d, "t2.fq")
    # the pairs a slab at a time (the generator's index arrays are 8 bytes per base); the mate files are the slabs' files joined,
    # labels counted through
    slab, b1, b2 = 2_000_000, 0, 0
    rec = (2 + 8 + 1) + L + 3 + L + 1
    with open(fq1, "wb") as f1, open(fq2, "wb") as f2:
        for lo in range(0, npairs, slab):
            n = min(slab, npairs - lo)
            pairs = make_pairs_torch(torch, 778 + lo // slab, d_seq, seq_lengths, seq_offsets, n, L, 0.01, 0.02, device).cpu().numpy().reshape(n, 2, L)
            for f, side in ((f1, 0), (f2, 1)):
                a = np.empty((n, rec), dtype=np.uint8)
                a[:, 0] = ord("@"); a[:, 1] = ord("r")
                idx = np.arange(lo, lo + n, dtype=np.int64)
                for k in range(8):
                    a[:, 2 + k] = ((idx // 10 ** (7 - k)) % 10 + ord("0")).astype(np.uint8)
                a[:, 10] = ord("\n")
                a[:, 11:11 + L] = pairs[:, side, :]
                a[:, 11 + L] = ord("\n"); a[:, 12 + L] = ord("+"); a[:, 13 + L] = ord("\n")
                a[:, 14 + L:14 + 2 * L] = ord("I")
                a[:, rec - 1] = ord("\n")
                a.tofile(f)
            del pairs
    b1 = b2 = rec * npairs
    for src, head, tail in ((fq1, h1, t1), (fq2, h2, t2)):
        with open(src, "rb") as f:
            open(head, "wb").write(f.read(n_chk * rec))
            f.seek((npairs - n_chk) * rec)
            open(tail, "wb").write(f.read())
    runs = [watched(lambda: api.map_files(index, fq1, fq2, samout=sam, first_gpu=device.index, gpus=1, streams=E2E_STREAMS, batch=E2E_BATCH, cmdline="bench.py e2e pairs")) for _ in range(2)]
    rep = runs[-1]
    oi.map_file_pe(h1, h2, sam_o, threads=cores)
    want = [l for l in open(sam_o, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    same = sam_head_records(sam, len(want)) == want
    oi.map_file_pe(t1, t2, sam_o, threads=cores)
    want_tail = [l for l in open(sam_o, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    same_tail = len(want_tail) == 2 * n_chk and sam_tail_records(sam, 2 * n_chk) == want_tail
    out = {"value": round(rep["reads"] / rep["seconds"], 1), "unit": "reads/s", "reads": int(rep["reads"]), "pairs": int(npairs), "seconds": round(rep["seconds"], 3),
           "first_run_seconds": round(runs[0]["seconds"], 3),
           "what": f"urmapx_map_files (= urmap -map2): 2 x {b1 / 1e9:.2f} GB mate files -> {os.path.getsize(sam) / 1e9:.2f} GB SAM file in /dev/shm, "
                   f"{(b1 + b2) / 1e9:.2f} GB of FASTQ in all; both text stages on the device",
           "stage_busy_s": {k: round(rep[k], 3) for k in ("parse_s", "gpu_s", "format_s", "write_s")},
           "sam_records_identical_to_oracle": bool(same and same_tail), "sam_records_checked": len(want) + len(want_tail),
           "sam_slices_checked": {"head_pairs": [0, n_chk], "tail_pairs": [npairs - n_chk, npairs], "head_identical": bool(same), "tail_identical": bool(same_tail)},
           "lanes_view": lane_view(rep)}
    for p in (fq1, fq2, sam, h1, h2, t1, t2, sam_o):
        os.remove(p)
    return out


def kernel_table(api, pe, L, nb, kms, counters, total_bp, gather_loads_s, stage_ms=None, p3_ms=None, alone=None):
    """Per-launch roofline figures.  Algorithmic bytes per read from the reference algorithm's own access counts
    (SURVEY.md 8d), counted by the oracle on the sample: probe kernel 5 B per GetBlob + the read; search kernel 5 B per
    chain slot + compared reference bases + the result record; DP kernel the target bases of the DP windows + the
    query flanks they are aligned with.  Single-end: the search is six launches (main / DP / finalize, then the same
    for the few reads whose lists outgrew the first pass's); paired-end: one search kernel (seed + probe inside)."""
    c = counters
    probe_bytes = 5.0 * c["n_getblob"] + L
    if pe:
        # paired-end: seed + probe of a pair's mates run at the start of that pair inside the search kernel (round 3)
        rows = [("search_pe_kernel", float(kms[1]),
                 probe_bytes + 5.0 * c["n_rowhop"] + c["n_extbases"] + c["n_dptarget"] + api.RESULT_DTYPE.itemsize)]
    elif stage_ms is None:
        rows = [("seed_probe_kernel", float(kms[0]), probe_bytes)]
        rows.append(("search_se_kernel", float(kms[1]),
                     5.0 * c["n_rowhop"] + c["n_extbases"] + c["n_dptarget"] + api.RESULT_DTYPE.itemsize))
    else:
        # single-end: seed + probe run inside the search kernel (the next read's slots are gathered into LDS while the
        # current read is searched), so its algorithmic bytes are both stages'
        # round 5: with phase 3 parked the search stage is two launches of search_se_kernel (every read; then the reads parked at phase 3)
        # with phase 3's dp_kernel launch between them -- the search row is the two search launches, the DP row every dp_kernel launch
        p3_dp = float(p3_ms[1]) if p3_ms is not None and p3_ms[0] > 0 else 0.0
        rows = [("search_se_kernel", float(stage_ms[0]) - p3_dp, probe_bytes + 5.0 * c["n_rowhop"] + c["n_extbases"] + api.RESULT_DTYPE.itemsize)]
        rows.append(("dp_kernel", float(stage_ms[1]) + p3_dp, 2.0 * c["n_dptarget"]))
        rows.append(("finalize_se_kernel", float(stage_ms[2]), float(api.RESULT_DTYPE.itemsize)))
        rows.append(("second pass (search + dp + finalize over the reads whose lists outgrew the first)", float(sum(stage_ms[3:6])), 0.0))
        rows.append(("general kernel (reads outside the fast kernels' domain; usually none)", float(stage_ms[6]), 0.0))
    sector_peak = 64.0 * gather_loads_s / 1e9
    alone_ms = {}
    if alone is not None:  # the same launches with the device to themselves (run_timed)
        if pe:
            alone_ms["search_pe_kernel"] = float(alone["kms"][1])
        else:
            a3 = float(alone["p3_ms"][1]) if alone["p3_ms"][0] > 0 else 0.0
            alone_ms = {"search_se_kernel": float(alone["stage_ms"][0]) - a3, "dp_kernel": float(alone["stage_ms"][1]) + a3, "finalize_se_kernel": float(alone["stage_ms"][2])}
    kern = []
    for name, ev_ms, alg in rows:
        # Batches alternating over several contexts: the interval between the events around a launch on its context's stream begins when that STREAM is ready, not when
        # the kernel gets CUs -- it holds the wait for the other context's kernel to leave the device (pairs: 28.7 ms of events around a 20.0 ms kernel, rocprofv3
        # kernel trace of the same command, profiles/r6/kernel_stats_hg38scale_pe_two_contexts.csv).  A launch's duration is therefore taken from the one-context steps
        # that follow the timed region (run_timed: the same launch on the same batches with the device to itself; HIP events, this process, this run).
        ms = alone_ms[name] if name in alone_ms and alone_ms[name] > 0 else ev_ms
        ach = alg * nb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        k = {"kernel": name, "avg_ms": round(ms, 4), "alg_bytes_per_read": round(alg, 1),
             "achieved_GBs": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5)}
        if name in alone_ms and alone_ms[name] > 0:
            k["avg_ms_is"] = "the launch with the device to itself (one-context steps after the timed region)"
            k["timed_region_event_ms"] = round(ev_ms, 4)  # includes the wait for the other context's kernels
        src = {}
        t = pmc_traffic(name, nb, total_bp, L, "pe" if pe else "se", source=src)
        k["hbm_read_bytes_per_launch_pmc"] = t
        k["hbm_write_bytes_per_launch_pmc"] = pmc_traffic(name, nb, total_bp, L, "pe" if pe else "se", what="write", source=src)
        k["pmc_source"] = src or None  # the committed rocprofv3 --pmc pass the two figures are LOOKED UP in (same kernel, workload and code version), not counted in this run
        if name == "search_se_kernel" and p3_ms is not None and p3_ms[0] > 0:
            k["launches"] = {"first (seed + probe + phases 1-2 of every read; phases 4-5 of the reads with nothing to align in phase 3)": round(float(p3_ms[0]), 4),
                             "second (the reads parked at phase 3: replay of AlignHSP's bookkeeping, phases 4-5)": round(float(p3_ms[2]), 4)}
        if name == "dp_kernel" and p3_dp > 0:
            k["launches"] = {"phase 3's flank DPs": round(p3_dp, 4), "phase 6's (three rounds)": round(float(stage_ms[1]), 4)}
        if t and sector_peak > 0 and ms > 0:
            k["sector_GBs"] = round(t / (ms * 1e-3) / 1e9, 1)
            k["frac_of_random_gather_peak"] = round(k["sector_GBs"] / sector_peak, 4)
        kern.append(k)
    return kern


# what the SQ counters say limits each kernel (profiles/r2/pmc_sq_*.json)
KERNEL_LIMITER = {"seed_probe_kernel": "hbm random access (64 B sector per 5 B slot)",
                  "search_se_kernel": "instruction issue + memory latency (order-dependent schedule, one wavefront per read)",
                  "search_pe_kernel": "instruction issue + memory latency (order-dependent schedule, one wavefront per pair)",
                  "dp_kernel": "VALU / SALU instruction issue (fp32 banded DP, no MFMA: max-plus recurrences)"}


def main():
    args = parse_args()
    from urmap_amd import ranks
    if args.gpus > 1 and not ranks.launched():
        # plain `python bench.py --gpus N`: start the N ranks as child processes (this process never touches the GPU)
        sys.exit(ranks.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    R = ranks.Ranks()
    rank, world = R.rank, R.world
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the mapping path has no CPU fallback")
    R.init(torch)
    dev_index = R.device_index
    device = torch.device("cuda", dev_index)

    from urmap_amd import api
    api.lib()  # fail loudly if the HIP library is missing

    t_setup = time.time()
    L = args.read_len
    total_bp = int(args.genome_mbp * 1e6)
    d_seq, seq_lengths, seq_offsets, labels, genome_desc = make_genome_torch(torch, 20260101, total_bp, device)
    slots, fasta_bytes = default_slot_count(seq_lengths, labels)
    t_gen = time.time() - t_setup
    index, blob_np, seq_np, d_seq, t_index = place_index(R, torch, api, device, d_seq, slots, seq_lengths, seq_offsets, labels)
    # UFIndex::Validate (ufindex.cpp:611-658) as a device pass over THIS rank's resident replica, before anything is timed: every
    # stored position re-hashed to its head slot, every chain followed, every used slot on exactly one chain
    index_ok, vrep = index.validate()
    if not index_ok:
        raise SystemExit(f"bench.py: rank {rank}: the resident index does not validate: {vrep}")
    # which genome and which table this run maps against (urmapx_index_checksum over the resident arrays), next to the values recorded in the
    # repo for this generator, seed and size (tests/golden/bench_genome.json: the store's, computed on the CPU; profiles/r6: the table's)
    table_checksum, genome_checksum = index.checksum()
    layout_checksums = index.layout_checksum()  # slot16, chain rows: derived from the table on this device at upload
    recorded = {}
    try:
        recorded = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_genome.json"))).get(f"{args.genome_mbp:g}", {})
    except (OSError, ValueError):
        pass
    contexts = max(1, args.contexts)
    mappers = [api.Mapper(index, device=dev_index, method=6) for _ in range(contexts if contexts > 1 else max(1, args.streams))]
    mapper = mappers[0]

    nb = args.reads_per_step
    pe = args.mode == "pe"
    n_batches = min(args.steps + args.warmup, 10)
    wl = Workload(torch, api, device, d_seq, seq_lengths, seq_offsets, pe, L, args.sub, args.indel, nb, n_batches, 1000 + 97 * rank,
                  streams=len(mappers), contexts=contexts)
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    # (the one-context steps behind the timed region run on rank 0 only: they feed the line's `alone_ms` / `sequential`, nothing is gathered from them)
    dt, kms, alone = run_timed(wl, mappers, args.steps, args.warmup, barrier=lambda: R.barrier(torch), alone_steps=5 if rank == 0 else 0)
    # every rank's own numbers, for the N > 1 line: its steps between the two barriers, its set-up (genome, index placement, batches)
    per_rank = R.all_gather_floats(torch, [wl.own_s, setup_s, t_index.get("make_ufi", 0.0), t_index.get("upload", 0.0), t_index.get("broadcast_s") or 0.0])
    dt = R.max_over_ranks(torch, dt)
    reads_per_s = world * args.steps * nb / dt

    leg_n = world > 1 and not pe and L == 150 and not args.no_e2e  # the N > 1 file-to-file leg: rank 0 maps over all N devices
    if leg_n and rank != 0:
        # this rank is done: its replica (122 GB at hg38 scale) and its contexts leave the device before rank 0 puts a replica of its
        # own there for the file-to-file leg (two resident indexes of 122 GB and the contexts of both would not fit 288 GB)
        for m in mappers:
            m.close()
        index.close()
        index._keep = ()
        del wl
        torch.cuda.empty_cache()
        R.barrier(torch)
    out = None
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        oi = ol.Index.wrap(24, 32, slots, blob_np, seq_np, seq_lengths, seq_offsets, labels)
        cores = host_cores()
        # parity on the last timed batch; the CPU baseline (N = 1 only) times the oracle on the same reads
        parity, counters, t_probe = wl.check(oi, 4000, cores)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            sample_n = int(min(nb, max(4000, 4000 * args.cpu_seconds / max(t_probe, 1e-3))))
            parity, counters, t_cpu = wl.check(oi, sample_n, cores)
            cpu_reads = parity["reads_checked"]
            ho = (np.arange(nb + 1, dtype=np.uint64) * L)
            extra = 0
            omap = (lambda bb: oi.map_pe(bb, ho, threads=cores)) if pe else (lambda bb: oi.map_se(bb, ho, threads=cores))
            while t_cpu < args.cpu_seconds * 0.6 and cpu_reads >= nb and extra + 1 < n_batches:
                hb2 = wl.batches[extra][: nb * L].cpu().numpy()
                t1 = time.perf_counter()
                omap(hb2)
                t_cpu += time.perf_counter() - t1
                cpu_reads += nb
                extra += 1
            cpu = {"value": round(cpu_reads / t_cpu, 1), "unit": "reads/s", "cores": cores, "host_logical_cpus": os.cpu_count(), "kind": "port",
                   "sample": f"{cpu_reads} reads of the timed batches (the last batch first), same index, "
                             f"oracle/liburmap_oracle.so (CPU restatement, SAM-identical to reference urmap) with {cores} "
                             f"OpenMP threads = the CPUs granted to this process ({os.cpu_count()} logical on the host), "
                             f"{t_cpu:.1f} s"}
            if cores > 10:  # the reference's own default thread count is min(cores, 10) (myutils.cpp:135-139)
                n10 = int(min(nb, max(4000, cpu_reads * 10 / cores / 3))) & ~1
                hb = wl.last[: n10 * L].cpu().numpy()
                h10 = (np.arange(n10 + 1, dtype=np.uint64) * L)
                t1 = time.perf_counter()
                (oi.map_pe if pe else oi.map_se)(hb, h10, threads=10)
                cpu["value_10_threads"] = round(n10 / (time.perf_counter() - t1), 1)
        elif world == 1:
            parity, counters, _ = wl.check(oi, min(nb, 200_000), cores)

        try:
            gather_loads_s = mapper.gather_microbench(1 << 28)
        except Exception:
            gather_loads_s = 0.0
        npl = nb if contexts > 1 else nb // len(mappers)  # reads per launch: whole batches alternate over the contexts, or (--contexts 1 --streams K) a batch is split
        kern = kernel_table(api, pe, L, npl, kms, counters, total_bp, gather_loads_s, None if pe else wl.stage_ms, None if pe else wl.p3_ms, alone=alone)
        dom = int(np.argmax([k["avg_ms"] if not k["kernel"].startswith(("second pass", "general kernel")) else 0.0 for k in kern]))
        dp_stats = wl.dp_stats
        key = "pe150" if pe else ("se150" if L == 150 else ("se250" if L == 250 else None))
        out = {
            "metric": metric_name(pe, L),
            "value": round(reads_per_s, 1),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/u64 (fp32 DP cells as the reference)",
            "data": "synthetic",
            "config": {"workload": f"{L} bp {'PE mates (pairs interleaved)' if pe else 'SE reads'} vs synthetic hg38-shaped {args.genome_mbp:g} Mbp genome "
                                   f"({slots} slots = GetPrime({fasta_bytes} FASTA bytes / 0.6), {5 * slots / 1e9:.2f} GB slot table + "
                                   f"{len(seq_np) / 1e9:.2f} GB sequence resident in HBM); {nb} reads/step, {args.sub:g} sub, {args.indel:g} indel",
                       "reads_per_step": nb, "contexts": contexts, "streams": 1 if contexts > 1 else len(mappers), "reads_per_launch": npl,
                       "read_len": L, "genome_bp": int(len(seq_np)), "slots": int(slots),
                       "genome": genome_desc,
                       "genome_checksum": f"{genome_checksum:016x}", "slot_table_checksum": f"{table_checksum:016x}",
                       "slot16_checksum": f"{layout_checksums[0]:016x}", "chain_rows_checksum": f"{layout_checksums[1]:016x}",
                       "genome_checksum_recorded": recorded.get("checksum"), "slot_table_checksum_recorded": recorded.get("slot_table_checksum"),
                       "inputs_are_the_recorded_ones": (None if not recorded.get("checksum") else
                                                        bool(recorded["checksum"] == f"{genome_checksum:016x}" and
                                                             recorded.get("slot_table_checksum", f"{table_checksum:016x}") == f"{table_checksum:016x}")),
                       "ranks": {"world": world, "backend": R.backend, "share_devices": bool(R.shared),
                                 "index_bytes_per_rank": t_index.get("index_bytes_per_rank"), "broadcast_s": t_index.get("broadcast_s"),
                                 "broadcast_pieces": t_index.get("broadcast_pieces")},
                       "index_validated": bool(index_ok),
                       "index_validation": {"what": "UFIndex::Validate (ufindex.cpp:611-658) as one device pass over the resident table, every rank its own replica: "
                                                    "each stored position re-hashed to its head slot, each chain followed link by link, each used slot on exactly one chain",
                                            "slots": int(vrep["slots"]), "used_slots": int(vrep["used"]), "rows": int(vrep["heads"]),
                                            "positions_rehashed": int(vrep["positions"]), "seconds": round(vrep["seconds"], 3)},
                       "setup_s": {"genome": round(t_gen, 1), **t_index, "total": round(setup_s, 1)}},
            # N > 1: what each rank did between the barriers (value is priced on the slowest) and how long it took to get ready
            "per_rank": {"ms_per_step": [round(1e3 * r[0] / args.steps, 3) for r in per_rank],
                         "ms_per_step_min": round(1e3 * min(r[0] for r in per_rank) / args.steps, 3),
                         "ms_per_step_max": round(1e3 * max(r[0] for r in per_rank) / args.steps, 3),
                         "slowest_rank": int(np.argmax([r[0] for r in per_rank])),
                         "setup_s": [round(r[1], 1) for r in per_rank],
                         "setup_parts_s": {"make_ufi (rank 0 builds, the others wait in the collective)": [round(r[2], 1) for r in per_rank],
                                           "upload / broadcast": [round(max(r[3], r[4]), 1) for r in per_rank]}},
            # bound: the limiter the counters show (profiles/r4: the waves of the search kernels wait on memory LATENCY more than half
            # of their cycles and issue in most of the rest; traffic is a few percent of the HBM peak) -- `peak` stays the HBM peak the
            # contract prices against, `frac` = algorithmic bytes / time / peak
            "roofline": {"bound": "latency+issue", "bound_in_contract_terms": "hbm", "peak_kind": "hbm", "kernel": kern[dom]["kernel"], "achieved": kern[dom]["achieved_GBs"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["frac"],
                         "limiter": KERNEL_LIMITER.get(kern[dom]["kernel"], ""),
                         "whole_step": {"alg_bytes_per_read": round(sum(k["alg_bytes_per_read"] for k in kern), 1),
                                        "achieved_GBs": round(sum(k["alg_bytes_per_read"] for k in kern) * nb / (dt / args.steps) / 1e9, 2)},
                         "duration_from": ("one-context steps after the timed region (`sequential`): HIP events around the launch with the device to itself, "
                                           f"{kern[dom]['avg_ms']} ms; in the timed region whole batches alternate over {contexts} contexts and the events around a launch also hold its wait "
                                           f"for the other context's kernels ({kern[dom]['timed_region_event_ms']} ms; rocprofv3's kernel trace of this command: profiles/r6/kernel_stats_hg38scale_*_two_contexts.csv)"
                                           if "timed_region_event_ms" in kern[dom] else "HIP events around the launch in the timed region (one context)"),
                         "traffic": kern[dom]["hbm_read_bytes_per_launch_pmc"],
                         "write_bytes": kern[dom]["hbm_write_bytes_per_launch_pmc"],
                         "traffic_source": ((kern[dom]["pmc_source"] or {}).get("fetch") and
                                            f"looked up, not counted in this run: {kern[dom]['pmc_source'].get('fetch')} / {kern[dom]['pmc_source'].get('write')} "
                                            f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel on this workload, code version {CODE_VERSION}); null when no committed pass names this kernel and version"),
                         "random_gather_peak": {"slot_reads_per_s": round(gather_loads_s), "sector_GBs": round(64.0 * gather_loads_s / 1e9, 1),
                                                "note": "measured in this run: independent random 5-byte slot reads over the resident table, 64 B sector each"}},
            "kernels": kern,
            "kernels_note": (f"whole batches alternate over {contexts} mapping contexts, each on its own HIP stream (how urmap -streams {contexts} runs a device): "
                             "avg_ms is a launch's duration with the device to itself (`sequential`: one context, steps back to back, after the timed region) -- those rows add up to sequential.ms_per_step; "
                             "timed_region_event_ms is the interval between the events around the launch on its context's stream in the timed region, which also holds the wait for the other context's kernels"
                             if contexts > 1 else
                             (f"a step = {len(mappers)} contexts x {nb // len(mappers)} reads on HIP streams of their own: avg_ms is per launch and launches of different contexts overlap, "
                              "so the rows add up to more than ms_per_step" if len(mappers) > 1 else "one context: the rows add up to ms_per_step")),
            "sequential": alone_block(alone, contexts),
            "parity": parity,
            "work_per_read": {k: round(v, 2) for k, v in counters.items()},
        }
        if dp_stats:
            out["phase6"] = {"hsps_given_to_dp_kernel": dp_stats[0], "reads_with_such_hsps": dp_stats[1], "dps_the_ordered_replay_used": dp_stats[2],
                             "dropped_by_a_round_gate_before_their_dp": dp_stats[3],
                             "second_pass": {"hsps": dp_stats[4], "reads": dp_stats[5], "used": dp_stats[6], "gated": dp_stats[7]}}
            if wl.round_ms is not None:
                out["phase6"]["launch_ms_by_round"] = {"rounds": mapper.dp_rounds(), "dp_kernel": [round(float(x[0]), 3) for x in wl.round_ms],
                                                       "finalize_se_kernel": [round(float(x[1]), 3) for x in wl.round_ms]}
        if not pe and wl.p3_ms[0] > 0:
            out["phase3"] = {"what": "Search_Lo's phase 3 (AlignHSP when the best HSP of phases 1-2 is long, search1m6.cpp:162-171) parked like phase 6: DpJobs for dp_kernel, "
                                     "the read resumed by a second launch of the search kernel from its parked state",
                             "reads_parked": int(round(wl.p3_stats[1])), "hsps_given_to_dp_kernel": int(round(wl.p3_stats[0])),
                             "launch_ms": {"search_first": round(float(wl.p3_ms[0]), 3), "dp_kernel": round(float(wl.p3_ms[1]), 3), "search_second": round(float(wl.p3_ms[2]), 3)}}
        if not pe and world == 1:
            # The probe stage alone (north_star: "rocprof HBM GB/s on the probe kernel reported against peak"): seed_probe_kernel launched on
            # its own over the last timed batch, OUTSIDE the timed region -- inside a mapping call the probe is a stage of search_se_kernel
            try:
                p0 = wl.parts[0]
                pms = [mapper.seed_probe_device(wl.last.data_ptr() + p0["lo"] * L, p0["d_offs"].data_ptr(), p0["n"], p0["n"] * L, L) for _ in range(3)][-1]
                palg = 5.0 * counters["n_getblob"] + L
                ptraffic = pmc_traffic("seed_probe_kernel", p0["n"], total_bp, L, "se")
                kmers = 2 * (L - 24 + 1)
                sect = kmers * 64.0 * 1.0625 + L  # one 64 B sector per k-mer, 6 % of the slots straddle two
                out["probe_only"] = {"kernel": "seed_probe_kernel", "ms": round(pms, 4), "reads": int(p0["n"]), "alg_bytes_per_read": round(palg, 1),
                                     "achieved_GBs": round(palg * p0["n"] / (pms * 1e-3) / 1e9, 1), "frac_of_hbm_peak": round(palg * p0["n"] / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                     "sector_bytes_per_read_model": round(sect, 1), "sector_GBs_model": round(sect * p0["n"] / (pms * 1e-3) / 1e9, 1),
                                     "hbm_read_bytes_per_launch_pmc": ptraffic,
                                     "sector_GBs_pmc": round(ptraffic / (pms * 1e-3) / 1e9, 1) if ptraffic else None,
                                     "frac_of_random_gather_peak": round((ptraffic if ptraffic else sect * p0["n"]) / (pms * 1e-3) / 1e9 / max(1e-9, 64.0 * gather_loads_s / 1e9), 4),
                                     "note": "launched standalone outside the timed region: SetSlotsVec + GetBlob for every k-mer of both strands, slots / tallies / positions written to HBM "
                                             "(13 B per k-mer, which the fused kernel keeps in LDS)"}
            except Exception as e:
                out["probe_only"] = {"error": str(e)[:200]}
        if key:
            out["work_per_read_survey"] = SURVEY_WORK_PER_READ[key]
        if cpu is not None:
            out["cpu_baseline"] = cpu

        # configs 3 and 5 on the same resident index (N = 1, headline run only): fewer steps, >= 100 k reads checked
        if world == 1 and not pe and L == 150 and not args.no_other_workloads:
            others = {}
            for name, (ope, oL, osub, oindel) in {"pe_2x150": (True, 150, 0.01, 0.001), "se_250_5pct": (False, 250, 0.04, 0.01)}.items():
                t0 = time.time()
                del wl
                torch.cuda.empty_cache()
                wl = Workload(torch, api, device, d_seq, seq_lengths, seq_offsets, ope, oL, osub, oindel, nb, 3, 5000, streams=len(mappers), contexts=contexts)
                osteps, owarm = 10, 2
                odt, okms, oalone = run_timed(wl, mappers, osteps, owarm)
                opar, ocnt, ot = wl.check(oi, min(nb, 200_000), cores)
                okern = kernel_table(api, ope, oL, npl, okms, ocnt, total_bp, gather_loads_s, None if ope else wl.stage_ms, None if ope else wl.p3_ms, alone=oalone)
                others[name] = {"metric": metric_name(ope, oL), "value": round(osteps * nb / odt, 1), "unit": "reads/s", "steps": osteps, "warmup": owarm,
                                "ms_per_step": round(1e3 * odt / osteps, 3), "contexts": contexts, "sequential": alone_block(oalone, contexts), "kernels": okern, "parity": opar,
                                "work_per_read": {k: round(v, 2) for k, v in ocnt.items()},
                                "cpu_port_reads_per_s": round(opar["reads_checked"] / ot, 1), "cpu_port_threads": cores,
                                "sub": osub, "indel": oindel, "wall_s": round(time.time() - t0, 1)}
                if wl.dp_stats:
                    ds = wl.dp_stats
                    others[name]["phase6"] = {"hsps_given_to_dp_kernel": ds[0], "reads_with_such_hsps": ds[1], "dps_the_ordered_replay_used": ds[2],
                                              "dropped_by_a_round_gate_before_their_dp": ds[3],
                                              "second_pass": {"hsps": ds[4], "reads": ds[5], "used": ds[6], "gated": ds[7]}}
                    if wl.round_ms is not None:
                        others[name]["phase6"]["launch_ms_by_round"] = {"rounds": mapper.dp_rounds(), "dp_kernel": [round(float(x[0]), 3) for x in wl.round_ms],
                                                                        "finalize_se_kernel": [round(float(x[1]), 3) for x in wl.round_ms]}
                if not ope and wl.p3_ms[0] > 0:
                    others[name]["phase3"] = {"reads_parked": int(round(wl.p3_stats[1])), "hsps_given_to_dp_kernel": int(round(wl.p3_stats[0])),
                                              "launch_ms": {"search_first": round(float(wl.p3_ms[0]), 3), "dp_kernel": round(float(wl.p3_ms[1]), 3),
                                                            "search_second": round(float(wl.p3_ms[2]), 3)}}
            out["other_workloads"] = others
        if world == 1 and not pe and L == 150 and not args.no_e2e:
            try:
                del wl
            except NameError:
                pass
            torch.cuda.empty_cache()
            def release_device():  # (the last leg runs the command line as a process of its own, with its own index on the device)
                for m in mappers:
                    m.close()
                index.close()
                index._keep = ()
                torch.cuda.empty_cache()
            out["e2e"] = run_e2e(torch, api, oi, index, device, d_seq, seq_lengths, seq_offsets, L, args.sub, args.indel,
                                 int(os.environ.get("URMAP_BENCH_E2E_READS", 10_000_000)), cores, ref_bin=ol.REF_BIN,  # BASELINE config 2: 10 M reads
                                 release_device=release_device)
            rb = out["e2e"].get("reference_binary")
            if rb and "cpu_baseline" in out:  # the reference itself, timed on this host in this run
                out["cpu_baseline"]["port_value"] = out["cpu_baseline"]["value"]
                out["cpu_baseline"]["value"] = rb["reads_per_s"]
                out["cpu_baseline"]["kind"] = "reference"
                out["cpu_baseline"]["compares_with"] = ("e2e.value: both are file to file (FASTQ text in, SAM text out) on the same FASTQ file; the kernel metric `value` "
                                                        "(reads resident in HBM, no text) compares with port_value, the CPU port on the same arrays")
                out["cpu_baseline"]["sample"] = (f"oracle/_ref/urmap (the unmodified reference, compiled by oracle/Makefile) -map -threads {rb['threads']} on the e2e FASTQ file: "
                                                 + rb["how"] + "; the CPU port on the same host: port_value (" + out["cpu_baseline"]["sample"] + ")")
        if leg_n:
            # the drop-in curve: urmap -map -gpus N file to file, run by rank 0 over all N devices (each gets its own replica of
            # the index; the other ranks wait at the barrier below) -- one writer feeds one SAM file whatever N is
            try:
                del wl
            except NameError:
                pass
            torch.cuda.empty_cache()
            R.barrier(torch)  # the other ranks have released their replicas and contexts
            if R.shared:
                os.environ["URMAPX_FORCE_DEVICE"] = str(dev_index)
            try:
                e2e_index = index
                if R.backend == "nccl" or os.environ.get("URMAP_BENCH_BROADCAST"):
                    # the ranks' replicas came over the broadcast and live in THEIR processes; rank 0's file-to-file run over all devices
                    # uploads its own from the host arrays (urmapx_index_replicate).  Rank 0's broadcast copy and its contexts are released
                    # first (the device held two full indexes for the length of this leg); d_seq stays: the reads are drawn from it
                    for m in mappers:
                        m.close()
                    index.close()
                    index._keep = ()  # the broadcast slot table (a torch tensor) goes with it
                    torch.cuda.empty_cache()
                    e2e_index = api.Index.wrap_host(24, 32, slots, blob_np, seq_np, seq_lengths, seq_offsets, labels).upload(dev_index)
                    out["config"]["ranks"]["e2e_index_chain_row_bytes"] = int(e2e_index.chain_row_bytes())
                out["e2e"] = run_e2e(torch, api, oi, e2e_index, device, d_seq, seq_lengths, seq_offsets, L, args.sub, args.indel,
                                     int(os.environ.get("URMAP_BENCH_E2E_READS", 12_500_000 * min(world, 8))), cores, ref_bin=None, gpus=world)  # config 4: 12.5 M reads per device
            except Exception as e:
                out["e2e"] = {"error": str(e)[:300]}
        print(json.dumps(out), flush=True)
    if world > 1 and not pe and L == 150 and not args.no_e2e:
        R.barrier(torch)
    R.close()


if __name__ == "__main__":
    main()
