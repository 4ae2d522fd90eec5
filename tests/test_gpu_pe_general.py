"""GPU: the general pair kernel (kernels_pe_slow.hip) -- State2::Search4 / Search5 with every list of both mates in global
memory, for the pairs the fast pair kernel flags (more than 1 024 hits on a mate, 8 192 HSPs, a path of more than 96 runs).
No pair the reference maps may make `urmap -map2` exit 1.

Two kinds of test:
  * URMAPX_TEST_PE_GENERAL=1 sends EVERY pair of a batch through the general kernel after the fast passes; the pair tests of
    test_gpu_parity.py are run again that way (oracle SAM, the reference's golden SAM and .tab, dense index with long links,
    rescue scan with hits, -veryfast, pair info), so the general kernel is held to everything the fast one is;
  * fixtures the fast kernel cannot take: mates with more than 1 024 live hits, and -- with relaxed penalties -- alignment
    paths of more than 96 runs, through the library and through `urmap -map2` on both text paths.
"""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
import test_gpu_parity as tp
import test_gpu_slow as ts
from conftest import reads_to_arrays
from test_gpu_parity import dense_case, rescue_case  # noqa: F401  (fixtures; small_case comes from conftest.py)
from urmap_amd import api, synth

pytestmark = pytest.mark.gpu
ACGT = np.frombuffer(b"ACGT", np.uint8)
EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "urmap_amd", "urmap")


@pytest.fixture
def general(monkeypatch):
    monkeypatch.setenv("URMAPX_TEST_PE_GENERAL", "1")


@pytest.mark.parametrize("rl,s1,s2,indel,n", [(150, 0.01, 0.02, 0.001, 1500), (120, 0.04, 0.08, 0.01, 1000), (270, 0.02, 0.04, 0.005, 400)])
def test_general_kernel_matches_oracle(general, small_case, tmp_path, rl, s1, s2, indel, n):
    tp.test_pe_matches_oracle(small_case, tmp_path, rl, s1, s2, indel, n)


@pytest.mark.parametrize("name", ["pe150", "pe100_noisy"])
def test_general_kernel_reproduces_reference_golden_sam(general, tmp_path, name):
    tp.test_pe_reproduces_reference_golden_sam(tmp_path, name)


def test_general_kernel_on_dense_index(general, dense_case, tmp_path):
    tp.test_pe_on_dense_index_walks_long_links(dense_case, tmp_path)


def test_general_kernel_rescue_scan(general, rescue_case):
    tp.test_pe_rescue_scan_produces_hits(rescue_case)


def test_general_kernel_veryfast(general, small_case, tmp_path):
    tp.test_pe_veryfast_search5_matches_oracle(small_case, tmp_path)


def test_general_kernel_pair_info(general, tmp_path):
    tp.test_pair_info_through_the_library(tmp_path)


@pytest.mark.parametrize("name,ufi_gz,with_sam", [("pe150", "g.ufi.gz", True), ("pe120_rep", "r.ufi.gz", True), ("pe120_rep", "r.ufi.gz", False)])
def test_general_kernel_cli_tabbedout(general, tmp_path, name, ufi_gz, with_sam):
    tp.test_cli_map2_tabbedout_reproduces_reference(tmp_path, name, ufi_gz, with_sam)


# ---------------------------------------------------------------------------------------------------------------------------
# what the fast pair kernel cannot take
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def huge_hits_case(tmp_path_factory):
    """test_gpu_slow.many_hits_case with 1 890 loci instead of 900: every 24-mer of the consensus survives in 30 loci (<= MaxIx
    32, so the index keeps it), each locus holds two of them -> reads with 1 000 .. 1 800 hits, beyond the pair kernel's third
    pass (1 024 per mate)."""
    d = str(tmp_path_factory.mktemp("hugehits"))
    rng = np.random.default_rng(12)
    rnd = lambda n: ACGT[rng.integers(0, 4, n)]
    C = rnd(200)
    parts = [rnd(5000)]
    for j in range(1890):
        m = C.copy()
        s = 20 + 2 * (j % 63)
        for p in list(range(s - 1, -1, -22)) + list(range(s + 25, 200, 22)):
            m[p] = ts._sub(rng, m[p])
        parts += [m, rnd(120 + int(rng.integers(0, 40)))]
    parts.append(rnd(5000))
    genome = [("chrH", np.concatenate(parts)), ("chrO", rnd(50000))]
    fa = os.path.join(d, "h.fa")
    synth.write_fasta(fa, genome)
    idx = ol.Index.build(fa, 2097169)
    ufi = os.path.join(d, "h.ufi")
    idx.save(ufi)
    reads = []
    for k, lo in enumerate((20, 10, 40, 0, 50, 30)):
        r = C[lo:lo + 150].copy()
        if k == 4:
            r[75] = ts._sub(rng, r[75])
        reads.append(r)
    return {"dir": d, "ufi": ufi, "oracle_index": idx, "reads": reads, "genome": genome}


def _huge_pairs(c):
    R = c["reads"]
    q = np.full(150, ord("I"), np.uint8)
    mates = [(R[0], synth.revcomp(R[4])), (R[2], synth.revcomp(R[0])), (synth.revcomp(R[1]), R[3]), (synth.revcomp(R[5]), synth.revcomp(R[2]))]
    pairs = []
    for k, (m1, m2) in enumerate(mates):
        pairs += [(f"c{k}/1", m1.copy(), q), (f"c{k}/2", m2.copy(), q)]
    r1, r2 = synth.make_pairs(17, c["genome"], 30, read_len=150, sub1=0.01, sub2=0.02)  # ordinary pairs in the same batch
    return pairs + [x for ab in zip(r1, r2) for x in ab]


def test_pairs_with_more_than_1024_hits_per_mate(huge_hits_case):
    c = huge_hits_case
    pairs = _huge_pairs(c)
    bases, offs = reads_to_arrays(pairs)
    ores, opaths, _ = c["oracle_index"].map_pe(bases, offs)
    assert (ores["hit_count"] > 1024).sum() >= 4, ores["hit_count"][:8]  # the fixture does what it was built for
    m = api.Mapper(api.Index.open(c["ufi"]).upload(0), device=0)
    g, gops = m.map_pe(bases, offs)
    ts._compare_pe(g, gops, ores, opaths)
    assert (np.minimum(ores["hit_count"], 0xFFFF) == g["hit_count"]).all()


def test_cli_map2_on_pairs_with_more_than_1024_hits(huge_hits_case, tmp_path):
    c = huge_hits_case
    pairs = _huge_pairs(c)
    r1 = [(lab[:-2] if lab.endswith(("/1", "/2")) else lab, s, q) for lab, s, q in pairs[0::2]]
    r2 = [(lab[:-2] if lab.endswith(("/1", "/2")) else lab, s, q) for lab, s, q in pairs[1::2]]
    f1, f2, sam, osam = (os.path.join(tmp_path, n) for n in ("r1.fq", "r2.fq", "out.sam", "oracle.sam"))
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    c["oracle_index"].map_file_pe(f1, f2, osam)
    want = [l for l in open(osam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    for env in ({}, {"URMAPX_HOST_TEXT": "1"}):
        rr = subprocess.run([EXE, "-map2", f1, "-reverse", f2, "-ufi", c["ufi"], "-samout", sam], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                            timeout=600, env={**os.environ, **env})
        assert rr.returncode == 0, rr.stderr.decode()[-2000:]
        assert [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")] == want


def _relaxed(P):
    """Penalties under which one-base gaps every few bases are cheaper than the mismatches of a shifted diagonal: the
    reference's constants are compile-time (state1.cpp:152-179); the library and the oracle take them as parameters."""
    P.max_penalty = 4000
    P.gap_open_score = -1
    P.gap_ext_score = -1
    return P


def _gappy(rng, src, clean, n_gaps, step=3):
    """`clean` bases as they are (the seed and its HSP), then n_gaps one-base gaps `step` bases apart, deletions and insertions
    in turn so that the alignment stays on the band"""
    out = list(src[:clean])
    p = clean
    for k in range(n_gaps):
        out += list(src[p:p + step])
        p += step
        if k % 2 == 0:
            p += 1  # a base of the reference the read does not have
        else:
            w = np.where(ACGT == src[p])[0]
            out.append(ACGT[(int(w[0]) + 1) % 4] if len(w) else ACGT[k % 4])  # a base the reference does not have
    return np.array(out, dtype=np.uint8)


def _nruns(path):
    return sum(1 for i in range(len(path)) if i == 0 or path[i] != path[i - 1])


def test_paths_of_more_than_96_runs(small_case, tmp_path):
    """With relaxed penalties a 2 kb single-end read (general single-end kernel) and 270-base mates (general pair kernel) get
    alignment paths of more than 96 runs: the path, and the CIGAR both formatters write for it, are the oracle's (the
    formatters used to cut a path at 96 runs)."""
    rng = np.random.default_rng(96)
    g0 = small_case["genome"][0][1]
    oi = small_case["oracle_index"]
    po, pg = _relaxed(ol.params(6)), _relaxed(api.params_for_method(6))
    m = api.Mapper(api.Index.open(small_case["ufi"]).upload(0), device=0, params=pg)
    # single-end: a third of the read clean (the HSP has to score 20 % of the read), then a one-base gap every three bases;
    # 600 / 1000 bases: the fast kernels flag the path, 2000 / 3000: the general kernel's own reads
    reads = []
    for k, QL in enumerate((1000, 1000, 2000, 2000, 600, 3000)):
        lo = int(rng.integers(1000, len(g0) - 6000))
        s = _gappy(rng, g0[lo:lo + 2 * QL], QL // 3, QL // 6)[:QL]
        reads.append((f"long{k}", s, np.full(len(s), ord("I"), np.uint8)))
    reads += synth.make_reads(7, small_case["genome"], 30, read_len=150, sub=0.02, ins=0.002, dele=0.002)
    bases, offs = reads_to_arrays(reads)
    ores, opaths, _ = oi.map_se(bases, offs, p=po)
    assert sum(_nruns(p) > 96 for p in opaths) >= 4, [_nruns(p) for p in opaths]
    g, gops = m.map_se(bases, offs)
    ts._compare(g, gops, ores, opaths)
    # pairs: 70 clean bases, then a one-base gap every three bases
    pairs = []
    for k in range(12):
        lo = int(rng.integers(1000, len(g0) - 2000))
        a = _gappy(rng, g0[lo:lo + 400], 70, 56)[:270]
        b = synth.revcomp(_gappy(rng, g0[lo + 150:lo + 550], 70, 56)[:270])
        pairs += [(f"p{k}/1", a, np.full(len(a), ord("I"), np.uint8)), (f"p{k}/2", b, np.full(len(b), ord("I"), np.uint8))]
    pairs += [x for ab in zip(*synth.make_pairs(5, small_case["genome"], 20, read_len=150, sub1=0.01, sub2=0.02)) for x in ab]
    bases, offs = reads_to_arrays(pairs)
    ores, opaths, _ = oi.map_pe(bases, offs, p=po)
    assert sum(_nruns(p) > 96 for p in opaths) >= 4, [_nruns(p) for p in opaths]
    g, gops = m.map_pe(bases, offs)
    ts._compare_pe(g, gops, ores, opaths)
