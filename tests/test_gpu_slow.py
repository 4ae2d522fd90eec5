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


@pytest.mark.parametrize("lens", [(1500, 2048, 3000), (8192,), (12000, 16000, 1025)])
def test_long_single_end_reads(small_case, lens):
    genome = small_case["genome"]
    reads = []
    for k, L in enumerate(lens):
        # the penalty cap is absolute (MAX_PENALTY 100, state1.cpp:152-179): a long read maps only if it is nearly exact
        reads += synth.make_reads(900 + k, genome, 3, read_len=L, sub=0.002, ins=0.0002, dele=0.0002, label_prefix=f"L{L}_")
    reads += synth.make_reads(77, genome, 40, read_len=150, sub=0.02, ins=0.002, dele=0.002)  # a mixed batch
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = small_case["oracle_index"].map_se(bases, offs)
    assert (ores["dbpos"][: 3 * len(lens)] != 0xFFFFFFFF).sum() >= len(lens), ores["dbpos"][: 3 * len(lens)]
    m = api.Mapper(api.Index.open(small_case["ufi"]).upload(0), device=0)
    g, gops = m.map_se(bases, offs)
    _compare(g, gops, ores, opaths)


def test_reads_beyond_the_general_kernel_are_flagged_not_mismapped(small_case):
    reads = synth.make_reads(5, small_case["genome"], 2, read_len=16001, sub=0.01) + synth.make_reads(6, small_case["genome"], 5, read_len=150)
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
