"""Seeded synthetic genomes and reads (numpy only; no reference code involved).

There is no hg38 (or any genome) on the build or GPU boxes, so every
test fixture and benchmark input comes from this generator.  Everything is a pure
function of the seed.

genome: uniform ACGT background, `repeat_frac` of it overwritten by copies of a few
repeat-family consensus sequences (divergence 0..max_div), optional N runs.
reads:  uniform start positions, 50 % minus strand, per-base substitution / insertion /
deletion rates, quality 'I'.
"""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.full(256, ord("N"), dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


def make_genome(seed: int, seq_lengths, repeat_frac=0.3, n_families=20, max_div=0.15,
                n_run_frac=0.02, label_prefix="chr"):
    """Return list of (label, uint8 array of ASCII bases)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fams = []
    for _ in range(n_families):
        flen = int(rng.integers(200, 3000))
        fams.append(ACGT[rng.integers(0, 4, size=flen)])
    out = []
    for si, L in enumerate(seq_lengths):
        seq = ACGT[rng.integers(0, 4, size=L)]
        # repeats
        covered = 0
        target = int(L * repeat_frac)
        while covered < target and n_families > 0:
            fam = fams[int(rng.integers(0, n_families))]
            flen = min(len(fam), L)
            if flen < 50:
                break
            pos = int(rng.integers(0, L - flen + 1))
            copy = fam[:flen].copy()
            div = rng.random() * max_div
            nmut = int(div * flen)
            if nmut:
                mpos = rng.integers(0, flen, size=nmut)
                copy[mpos] = ACGT[rng.integers(0, 4, size=nmut)]
            if rng.random() < 0.5:
                copy = revcomp(copy)
            seq[pos:pos + flen] = copy
            covered += flen
        # N runs
        nleft = int(L * n_run_frac)
        while nleft > 0:
            rl = int(min(nleft, rng.integers(10, 2000)))
            pos = int(rng.integers(0, max(1, L - rl)))
            seq[pos:pos + rl] = ord("N")
            nleft -= rl
        out.append((f"{label_prefix}{si + 1}", seq))
    return out


def write_fasta(path, genome, width=60, lowercase_frac=0.0, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    with open(path, "wb") as f:
        for label, seq in genome:
            f.write(b">" + label.encode() + b" synthetic\n")
            s = seq
            if lowercase_frac > 0:
                s = seq.copy()
                # soft-masked stretches, as in real assemblies
                n = int(len(s) * lowercase_frac / 500) + 1
                for _ in range(n):
                    p = int(rng.integers(0, max(1, len(s) - 500)))
                    s[p:p + 500] |= 0x20
            b = s.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width])
                f.write(b"\n")


def _mutate(rng, frag: np.ndarray, sub, ins, dele):
    if sub == 0 and ins == 0 and dele == 0:
        return frag
    out = []
    r = rng.random(size=len(frag))
    pick = rng.integers(0, 3, size=len(frag))
    for i, c in enumerate(frag):
        x = r[i]
        if x < sub:
            # substitute by one of the 3 other letters (N stays N)
            idx = {65: 0, 67: 1, 71: 2, 84: 3}.get(int(c))
            if idx is None:
                out.append(c)
            else:
                out.append(ACGT[(idx + 1 + pick[i]) % 4])
        elif x < sub + ins:
            out.append(c)
            out.append(ACGT[pick[i]])
        elif x < sub + ins + dele:
            continue
        else:
            out.append(c)
    return np.array(out, dtype=np.uint8)


def make_reads(seed: int, genome, n, read_len=150, sub=0.01, ins=0.0005, dele=0.0005,
               random_frac=0.0, label_prefix="r"):
    """Return list of (label, seq uint8, qual uint8)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([len(s) for _, s in genome], dtype=np.int64)
    cum = np.cumsum(lens)
    reads = []
    for i in range(n):
        if rng.random() < random_frac:
            seq = ACGT[rng.integers(0, 4, size=read_len)]
            reads.append((f"{label_prefix}{i}_random", seq, np.full(read_len, ord("I"), np.uint8)))
            continue
        while True:
            g = int(rng.integers(0, cum[-1]))
            si = int(np.searchsorted(cum, g, side="right"))
            L = int(lens[si])
            span = read_len + 16
            if L <= span:
                continue
            pos = int(rng.integers(0, L - span))
            break
        frag = genome[si][1][pos:pos + span]
        minus = rng.random() < 0.5
        m = _mutate(rng, frag, sub, ins, dele)[:read_len]
        if len(m) < read_len:
            continue
        if minus:
            m = revcomp(m)
        label = f"{label_prefix}{i}_{genome[si][0]}_{pos + 1}_{'-' if minus else '+'}"
        reads.append((label, m, np.full(read_len, ord("I"), np.uint8)))
    return reads


def make_pairs(seed: int, genome, n, read_len=150, insert_mean=300, insert_sd=50,
               sub1=0.01, sub2=0.015, ins=0.0005, dele=0.0005, label_prefix="p"):
    """Return (reads1, reads2): FR-oriented pairs with '/1' '/2' labels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.array([len(s) for _, s in genome], dtype=np.int64)
    cum = np.cumsum(lens)
    r1, r2 = [], []
    q = np.full(read_len, ord("I"), np.uint8)
    i = 0
    while len(r1) < n:
        g = int(rng.integers(0, cum[-1]))
        si = int(np.searchsorted(cum, g, side="right"))
        L = int(lens[si])
        isz = int(max(read_len + 20, rng.normal(insert_mean, insert_sd)))
        if L <= isz + 40:
            continue
        pos = int(rng.integers(0, L - isz - 40))
        frag = genome[si][1][pos:pos + isz + 32]
        a = _mutate(rng, frag[:read_len + 16], sub1, ins, dele)[:read_len]
        b = _mutate(rng, revcomp(frag[:isz])[:read_len + 16], sub2, ins, dele)[:read_len]
        if len(a) < read_len or len(b) < read_len:
            continue
        if rng.random() < 0.5:
            a, b = b, a
        base = f"{label_prefix}{i}_{genome[si][0]}_{pos + 1}_{isz}"
        r1.append((base + "/1", a, q))
        r2.append((base + "/2", b, q))
        i += 1
    return r1, r2


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for label, seq, qual in reads:
            f.write(b"@" + label.encode() + b"\n")
            f.write(seq.tobytes())
            f.write(b"\n+\n")
            f.write(qual.tobytes())
            f.write(b"\n")
