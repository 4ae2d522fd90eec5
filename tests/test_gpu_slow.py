"""GPU: reads outside the fast kernels' domain are mapped by the general kernel (kernels_slow.hip), bit-identical to the
oracle -- no read the reference maps may make `urmap -map` exit 1.

  * many hits: the reference's hit list grows without bound (state1.cpp:193-228).  An index drops every k-mer with more
    than MaxIx = 32 occurrences (ufindex.cpp UpdateSlot), so a plain tandem satellite gives a read at most ~34 hits; the
    fixture below is built so that each of a read's k-mers survives in a DIFFERENT small set of near-copies: 900 loci,
    each holding exactly two of the read's k-mers, every k-mer in <= 32 loci -> 450..820 hits per read (> 512, the fast
    kernels' second-pass capacity).
  * long reads: 1.5 kb .. 12 kb single-end reads (the reference's scratch holds ~30 kb, state1.h:113).
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from conftest import reads_to_arrays
from urmap_amd import api, synth

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", np.uint8)


def _sub(rng, base):
    return ACGT[(int(np.where(ACGT == base)[0][0]) + 1 + int(rng.integers(0, 3))) % 4]


@pytest.fixture(scope="module")
def many_hits_case(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("manyhits"))
    rng = np.random.default_rng(11)
    rnd = lambda n: ACGT[rng.integers(0, 4, n)]
    C = rnd(200)
    parts = [rnd(5000)]
    for j in range(900):
        m = C.copy()
        s = 20 + 2 * (j % 63)  # the only stretch of 25 unchanged bases: it holds two 24-mers of the consensus
        for p in list(range(s - 1, -1, -22)) + list(range(s + 25, 200, 22)):
            m[p] = _sub(rng, m[p])
        parts += [m, rnd(120 + int(rng.integers(0, 40)))]
    parts.append(rnd(5000))
    genome = [("chrH", np.concatenate(parts)), ("chrO", rnd(50000))]
    fa = os.path.join(d, "h.fa")
    synth.write_fasta(fa, genome)
    idx = ol.Index.build(fa, 1048583)
    ufi = os.path.join(d, "h.ufi")
    idx.save(ufi)
    reads = []
    for k, lo in enumerate((20, 10, 40, 0, 50, 30)):
        r = C[lo:lo + 150].copy()
        if k == 4:
            r[75] = _sub(rng, r[75])
        if k % 2:
            r = synth.revcomp(r)
        reads.append((f"h{k}", r, np.full(150, ord("I"), np.uint8)))
    return {"dir": d, "ufi": ufi, "oracle_index": idx, "reads": reads, "genome": genome}


def _compare(g, gops, ores, opaths, min_fields=("dbpos", "seq_index", "coord", "score", "second", "mapq", "exit_phase")):
    assert (g["status"] == 0).all(), np.unique(g["status"])
    for name in min_fields:
        assert (g[name].astype(np.int64) == ores[name].astype(np.int64)).all(), name
    assert (np.minimum(ores["hit_count"], 0xFFFF) == g["hit_count"]).all()
    mapped = ores["dbpos"] != 0xFFFFFFFF
    assert (g["plus"][mapped] == ores["plus"][mapped]).all()
    for i in np.nonzero(mapped)[0]:
        o = int(g["path_off"][i])
        assert api.decode_path(gops[o:o + int(g["path_nops"][i])]) == opaths[i], i


def test_reads_with_more_than_512_hits(many_hits_case):
    c = many_hits_case
    bases, offs = reads_to_arrays(c["reads"])
    ores, opaths, _ = c["oracle_index"].map_se(bases, offs)
    assert (ores["hit_count"] > 512).sum() >= 3, ores["hit_count"]  # the fixture does what it was built for
    m = api.Mapper(api.Index.open(c["ufi"]).upload(0), device=0)
    g, gops = m.map_se(bases, offs)
    _compare(g, gops, ores, opaths)


@pytest.mark.parametrize("lens", [(1500, 2048, 3000), (8192,), (12000, 16000, 1025), (25000, 30836)])  # 30 836: the reference's own limit (state1.h:113)
def test_long_single_end_reads(small_case, lens):
    genome = small_case["genome"]
    reads = []
    for k, L in enumerate(lens):
        # the penalty cap is absolute (MAX_PENALTY 100, state1.cpp:152-179): a long read maps only if it is nearly exact
        reads += synth.make_reads(900 + k, genome, 3, read_len=L, sub=min(0.002, 15.0 / L), ins=min(0.0002, 1.5 / L), dele=min(0.0002, 1.5 / L), label_prefix=f"L{L}_")
    reads += synth.make_reads(77, genome, 40, read_len=150, sub=0.02, ins=0.002, dele=0.002)  # a mixed batch
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = small_case["oracle_index"].map_se(bases, offs)
    assert (ores["dbpos"][: 3 * len(lens)] != 0xFFFFFFFF).sum() >= len(lens), ores["dbpos"][: 3 * len(lens)]
    m = api.Mapper(api.Index.open(small_case["ufi"]).upload(0), device=0)
    g, gops = m.map_se(bases, offs)
    _compare(g, gops, ores, opaths)


def test_reads_beyond_the_general_kernel_are_flagged_not_mismapped(small_case):
    reads = synth.make_reads(5, small_case["genome"], 2, read_len=30837, sub=0.01) + synth.make_reads(6, small_case["genome"], 5, read_len=150)
    bases, offs = reads_to_arrays(reads)
    m = api.Mapper(api.Index.open(small_case["ufi"]).upload(0), device=0)
    g, _ = m.map_se(bases, offs, allow_unsupported=True)
    assert (g["status"][:2] == 0x10).all() and (g["status"][2:] == 0).all()


def test_cli_maps_what_the_fast_kernels_flag(many_hits_case, small_case, tmp_path):
    """`urmap -map` exits 0 and writes the oracle's records for reads with > 512 hits and for 2 kb / 8 kb reads."""
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urmap_amd", "urmap")
    for case, reads in ((many_hits_case, many_hits_case["reads"] + synth.make_reads(3, many_hits_case["genome"], 60, read_len=150, sub=0.02)),
                        (small_case, synth.make_reads(31, small_case["genome"], 4, read_len=2048, sub=0.002, ins=0.0002, dele=0.0002)
                         + synth.make_reads(32, small_case["genome"], 2, read_len=8192, sub=0.001, ins=0.0001, dele=0.0001)
                         + synth.make_reads(33, small_case["genome"], 100, read_len=150, sub=0.02))):
        fq, sam, osam = (os.path.join(tmp_path, n) for n in ("r.fq", "out.sam", "oracle.sam"))
        synth.write_fastq(fq, reads)
        case["oracle_index"].map_file_se(fq, osam)
        want = [l for l in open(osam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        for extra in ([], ["-batch", "16"]):
            for env in ({}, {"URMAPX_HOST_TEXT": "1"}):
                r = subprocess.run([exe, "-map", fq, "-ufi", case["ufi"], "-samout", sam] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   timeout=300, env={**os.environ, **env})
                assert r.returncode == 0, r.stderr.decode()[-2000:]
                got = [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
                assert got == want


def _compare_pe(g, gops, ores, opaths):
    assert (g["status"] == 0).all(), np.unique(g["status"])
    for name in ("dbpos", "seq_index", "coord", "score", "second", "mapq"):
        assert (g[name].astype(np.int64) == ores[name].astype(np.int64)).all(), name
    mapped = ores["dbpos"] != 0xFFFFFFFF
    assert (g["plus"][mapped] == ores["plus"][mapped]).all()
    for i in np.nonzero(mapped)[0]:
        o = int(g["path_off"][i])
        assert api.decode_path(gops[o:o + int(g["path_nops"][i])]) == opaths[i], i


def test_pairs_with_more_than_256_hits_per_mate(many_hits_case):
    """A mate with more than 256 live hits outgrows the pair kernel's second pass; a third pass with lists of 1024 hits per
    mate maps it (the reference's lists have no bound, state1.cpp:193-228).  FindPairs sees hundreds x hundreds of hits:
    the best and second-best pair are tracked without a pair list (state2.cpp:20-85 keeps one of unbounded length)."""
    c = many_hits_case
    R = [x[1] for x in c["reads"]]  # six cuts of the consensus; the odd-numbered ones are stored reverse-complemented
    q = np.full(150, ord("I"), np.uint8)
    mates = [(R[0], synth.revcomp(R[4])), (R[2], synth.revcomp(R[0])), (synth.revcomp(R[1]), R[3]), (synth.revcomp(R[5]), synth.revcomp(R[2]))]
    pairs = []
    for k, (m1, m2) in enumerate(mates):
        pairs += [(f"c{k}/1", m1.copy(), q), (f"c{k}/2", m2.copy(), q)]
    r1, r2 = synth.make_pairs(17, c["genome"], 30, read_len=150, sub1=0.01, sub2=0.02)  # ordinary pairs in the same batch
    pairs += [x for ab in zip(r1, r2) for x in ab]
    bases, offs = reads_to_arrays(pairs)
    ores, opaths, _ = c["oracle_index"].map_pe(bases, offs)
    assert (ores["hit_count"] > 256).sum() >= 4, ores["hit_count"]  # the fixture does what it was built for
    m = api.Mapper(api.Index.open(c["ufi"]).upload(0), device=0)
    g, gops = m.map_pe(bases, offs)
    _compare_pe(g, gops, ores, opaths)


@pytest.fixture(scope="module")
def satellite_case(tmp_path_factory):
    """VERDICT r2 item 3's fixture, literally: 2 000 tandem copies of a 171-base monomer, each copy <= 2 % diverged, between
    random flanks.  (An index drops every k-mer with more than 32 occurrences, so reads from it end with a few dozen hits.)"""
    d = str(tmp_path_factory.mktemp("satellite"))
    rng = np.random.default_rng(171)
    rnd = lambda n: ACGT[rng.integers(0, 4, n)]
    mono = rnd(171)
    copies = []
    for _ in range(2000):
        m = mono.copy()
        for p in rng.choice(171, int(rng.integers(0, 4)), replace=False):  # 0..3 substitutions = <= 1.75 %
            m[p] = _sub(rng, m[p])
        copies.append(m)
    genome = [("chrS", np.concatenate([rnd(20000)] + copies + [rnd(20000)])), ("chrO", rnd(60000))]
    fa = os.path.join(d, "s.fa")
    synth.write_fasta(fa, genome)
    idx = ol.Index.build(fa, 2097169)
    ufi = os.path.join(d, "s.ufi")
    idx.save(ufi)
    return {"dir": d, "ufi": ufi, "oracle_index": idx, "genome": genome}


def test_tandem_satellite_single_and_paired(satellite_case):
    c = satellite_case
    sat = [("chrS", c["genome"][0][1][20000 - 300:20000 + 171 * 2000 + 300])]
    reads = synth.make_reads(61, sat, 300, read_len=150, sub=0.01, ins=0.001, dele=0.001) + synth.make_reads(62, c["genome"], 100, read_len=150, sub=0.02)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = c["oracle_index"].map_se(bases, offs)
    m = api.Mapper(api.Index.open(c["ufi"]).upload(0), device=0)
    g, gops = m.map_se(bases, offs)
    _compare(g, gops, ores, opaths)
    r1, r2 = synth.make_pairs(63, sat, 200, read_len=150, sub1=0.01, sub2=0.02, ins=0.001, dele=0.001)
    pairs = [x for ab in zip(r1, r2) for x in ab]
    bases, offs = reads_to_arrays(pairs)
    ores, opaths, _ = c["oracle_index"].map_pe(bases, offs)
    g, gops = m.map_pe(bases, offs)
    _compare_pe(g, gops, ores, opaths)


def test_cli_map2_on_pairs_with_hundreds_of_hits(many_hits_case, tmp_path):
    c = many_hits_case
    r = [c["reads"][k][1] if k % 2 == 0 else synth.revcomp(c["reads"][k][1]) for k in (0, 2, 3)]
    q = np.full(150, ord("I"), np.uint8)
    r1 = [(f"c{k}", x.copy(), q) for k, x in enumerate(r)] + synth.make_reads(8, c["genome"], 40, read_len=150, sub=0.02)
    r2 = [(f"c{k}", synth.revcomp(x), q) for k, x in enumerate(r)] + synth.make_reads(9, c["genome"], 40, read_len=150, sub=0.02)
    f1, f2, sam, osam = (os.path.join(tmp_path, n) for n in ("r1.fq", "r2.fq", "out.sam", "oracle.sam"))
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    c["oracle_index"].map_file_pe(f1, f2, osam)
    want = [l for l in open(osam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urmap_amd", "urmap")
    for env in ({}, {"URMAPX_HOST_TEXT": "1"}):
        rr = subprocess.run([exe, "-map2", f1, "-reverse", f2, "-ufi", c["ufi"], "-samout", sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            timeout=300, env={**os.environ, **env})
        assert rr.returncode == 0, rr.stderr.decode()[-2000:]
        assert [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")] == want
