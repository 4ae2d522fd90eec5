"""CPU: the C-ABI library loads and exports every symbol include/urmapx.h declares; the host-side pieces
(index file parsing, -make_ufi, SAM text) are checked against the golden fixtures and the oracle.  No compute
entry point is called here -- those need a GPU and fail loudly without one."""
import ctypes as C
import filecmp
import gzip
import os
import re

import numpy as np
import pytest

import oracle_lib as ol
from urmap_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gold_ufi(tmp_path_factory):
    d = tmp_path_factory.mktemp("gold")
    p = os.path.join(d, "g.ufi")
    with gzip.open(os.path.join(GOLD, "g.ufi.gz"), "rb") as z, open(p, "wb") as f:
        f.write(z.read())
    return p


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "urmapx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(urmapx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = api.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)


def test_params_match_setmethod():
    """State1::SetMethod constants, state1.cpp:152-179."""
    p6, p7 = api.params_for_method(6), api.params_for_method(7)
    assert (p6.mismatch_score, p6.gap_open_score, p6.gap_ext_score, p6.xdrop, p6.max_penalty, p6.band_radius) == (-3, -5, -1, 9, 100, 12)
    assert (p7.mismatch_score, p7.gap_open_score, p7.gap_ext_score, p7.xdrop, p7.max_penalty, p7.band_radius) == (-4, -6, -2, 12, 75, 8)
    o6 = ol.params(6)
    for f, _ in api.Params._fields_:
        assert getattr(p6, f) == getattr(o6, f), f


def test_index_open_parses_reference_ufi(gold_ufi):
    """UFIndex::FromFile, ufindexio.cpp:60-115."""
    idx = api.Index.open(gold_ufi)
    assert (idx.word_length, idx.max_ix) == (24, 32)
    assert idx.directory() == [("chr1", 22000, 0), ("chr2", 13000, 22032), ("chr3", 5000, 35064)]
    w, maxix, sds, slots = ol.ufi_header(gold_ufi)
    assert (idx.slot_count, idx.seqdata_size) == (slots, sds)
    assert idx.sam_header_sq() == b"@SQ\tSN:chr1\tLN:22000\n@SQ\tSN:chr2\tLN:13000\n@SQ\tSN:chr3\tLN:5000\n"


def test_index_open_rejects_garbage(tmp_path):
    p = os.path.join(tmp_path, "bad.ufi")
    open(p, "wb").write(b"not an index at all")
    with pytest.raises(api.UrmapxError) as e:
        api.Index.open(p)
    assert e.value.code == -2
    with pytest.raises(api.UrmapxError) as e:
        api.Index.open(os.path.join(tmp_path, "missing.ufi"))
    assert e.value.code == -1


def test_product_make_ufi_is_byte_identical(gold_ufi, tmp_path):
    """cmd_make_ufi (ufindexio.cpp:117-179) in the product library vs the reference's own .ufi."""
    w, maxix, sds, slots = ol.ufi_header(gold_ufi)
    out = os.path.join(tmp_path, "p.ufi")
    api.make_ufi(os.path.join(GOLD, "g.fa"), out, slots)
    assert filecmp.cmp(out, gold_ufi, shallow=False)


def test_compute_entry_points_fail_loudly_without_gpu(gold_ufi):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    idx = api.Index.open(gold_ufi)
    with pytest.raises(api.UrmapxError):
        idx.upload(0)


def oracle_results_as_product(ores, opaths):
    """Re-express the oracle's per-read results in the product's result/arena layout (test glue)."""
    res = np.zeros(len(ores), dtype=api.RESULT_DTYPE)
    ops = []
    for i, r in enumerate(ores):
        for f in ("dbpos", "seq_index", "coord", "score", "second", "mapq", "plus", "exit_phase", "hit_count"):
            res[f][i] = r[f]
        p = opaths[i]
        if r["dbpos"] != 0xFFFFFFFF and p:
            res["path_off"][i] = len(ops)
            n0 = len(ops)
            k = 0
            while k < len(p):
                j = k
                while j < len(p) and p[j] == p[k]:
                    j += 1
                ops.append(((j - k) << 2) | "MDI".index(p[k]))
                k = j
            res["path_nops"][i] = len(ops) - n0
    return res, np.array(ops, dtype=np.uint16)


@pytest.mark.parametrize("name", ["se150", "se250", "se_short"])
def test_product_sam_text_equals_reference_golden(gold_ufi, name):
    """SetSAM / GetCIGAR / PathToCIGAR / CIGAROpsFixDanglingMs (setsam.cpp, cigar.cpp) in the product's host code:
    fed with the oracle's hits it must reproduce the reference's SAM records byte for byte."""
    labels, bases, offs, quals = api.read_fastq_arrays(os.path.join(GOLD, name + ".fq"))
    oi = ol.Index.load(gold_ufi)
    ores, opaths, _ = oi.map_se(bases, offs)
    res, ops = oracle_results_as_product(ores, opaths)
    idx = api.Index.open(gold_ufi)
    sam = idx.sam_header_sq() + idx.sam_se(res, ops, labels, bases, offs, quals)
    want = open(os.path.join(GOLD, name + ".sam"), "rb").read()
    assert sam == want


@pytest.mark.parametrize("name,ufi_gz", [("pe150", "g.ufi.gz"), ("pe100_noisy", "g.ufi.gz"), ("pe120_rep", "r.ufi.gz")])
def test_product_pe_sam_text_equals_reference_golden(tmp_path, name, ufi_gz):
    """State2::SetSAM2 / GetPairedFlags / TLEN (output2.cpp:18-132) in the product's host code (urmapx_sam_pe): fed
    with the oracle's per-mate results it must reproduce the reference's -map2 SAM records byte for byte."""
    import gzip
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(GOLD, ufi_gz), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    labels, bases, offs, quals = api.interleave_pairs(api.read_fastq_arrays(os.path.join(GOLD, name + "_1.fq")),
                                                      api.read_fastq_arrays(os.path.join(GOLD, name + "_2.fq")))
    ores, opaths, _ = ol.Index.load(ufi).map_pe(bases, offs)
    res, ops = oracle_results_as_product(ores, opaths)
    idx = api.Index.open(ufi)
    sam = idx.sam_header_sq() + idx.sam_pe(res, ops, labels, bases, offs, quals)
    want = open(os.path.join(GOLD, name + ".sam"), "rb").read()
    assert sam == want


@pytest.mark.parametrize("name,ufi_gz,sam_on", [("pe150", "g.ufi.gz", True), ("pe100_noisy", "g.ufi.gz", True),
                                                ("pe120_rep", "r.ufi.gz", True), ("pe120_rep", "r.ufi.gz", False)])
def test_product_tabbedout_text_equals_reference_golden(tmp_path, name, ufi_gz, sam_on):
    """State2::OutputTab2 (outputtab2.cpp:85-120) in the product's host code (urmapx_tab_pe): fed with the oracle's
    per-mate results and pair records it must reproduce the reference's -tabbedout file byte for byte, with and
    without SAM output switched on."""
    import gzip
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(GOLD, ufi_gz), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    labels, bases, offs, quals = api.interleave_pairs(api.read_fastq_arrays(os.path.join(GOLD, name + "_1.fq")),
                                                      api.read_fastq_arrays(os.path.join(GOLD, name + "_2.fq")))
    ores, opaths, oinfo = ol.Index.load(ufi).map_pe_info(bases, offs)
    res, _ = oracle_results_as_product(ores, opaths)
    info = np.zeros(len(oinfo), dtype=api.PAIR_INFO_DTYPE)
    for f in info.dtype.names:
        info[f] = oinfo[f]
    tab = api.Index.open(ufi).tab_pe(res, info, labels, offs, sam_on=sam_on)
    want = open(os.path.join(GOLD, name + (".tab" if sam_on else "_nosam.tab")), "rb").read()
    assert tab == want


def test_cigar_dangling_m_rules():
    """cigar.cpp:141-199: a terminal M of <= 2 next to an indel > 4 is merged into the M beyond it."""
    idx_path = None
    # exercised through urmapx_sam_se with hand-made paths on a one-sequence dummy index
    seq = np.frombuffer(b"ACGT" * 100, dtype=np.uint8)
    blob = np.zeros(5 * 101 + 8, dtype=np.uint8)
    idx = api.Index.wrap_host(24, 32, 101, blob, seq, [400], [0], ["c"])

    def cigar(path, L):
        res = np.zeros(1, dtype=api.RESULT_DTYPE)
        res["dbpos"] = 0; res["seq_index"] = 0; res["coord"] = 0; res["plus"] = 1
        _, ops = oracle_results_as_product(
            np.array([(0, 0, 0, 0, 0, 0, 0, 0, 1, 6, len(path), 0)], dtype=ol.RESULT_DTYPE), [path])
        res["path_nops"] = len(ops)
        rec = idx.sam_se(res, ops, ["r"], np.full(L, 65, np.uint8), np.array([0, L], np.uint64), np.full(L, 73, np.uint8))
        return rec.split(b"\t")[5].decode()

    assert cigar("M" * 1 + "D" * 6 + "M" * 100, 107) == "6I101M"      # path D = read-only base = CIGAR I
    assert cigar("M" * 100 + "I" * 6 + "M" * 2, 102) == "102M6D"
    assert cigar("M" * 3 + "D" * 6 + "M" * 100, 109) == "3M6I100M"    # 3 > 2: untouched
    assert cigar("M" * 2 + "D" * 4 + "M" * 100, 106) == "2M4I100M"    # indel not > 4: untouched
    assert cigar("M" * 1 + "D" * 6 + "M" * 100 + "I" * 6 + "M" * 1, 108) == "6I101M6D1M"  # head rule XOR tail rule
    assert cigar("", 150) == "150M"


def test_cli_make_ufi_is_byte_identical(gold_ufi, tmp_path):
    """`urmap -make_ufi FASTA -output UFI -slots N` of this build vs the reference's .ufi (no GPU involved)."""
    import subprocess
    exe = os.path.join(ROOT, "urmap_amd", "urmap")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    w, maxix, sds, slots = ol.ufi_header(gold_ufi)
    out = os.path.join(tmp_path, "cli.ufi")
    r = subprocess.run([exe, "-make_ufi", os.path.join(GOLD, "g.fa"), "-output", out, "-slots", str(slots)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert filecmp.cmp(out, gold_ufi, shallow=False)
    # without -slots: GetPrime(file size / 0.6) (ufindexio.cpp:138-150, prime.cpp:11-21) -- the fixture was written by
    # the reference with its default slot count, so the default-sized index must be that file too
    r = subprocess.run([exe, "-make_ufi", os.path.join(GOLD, "g.fa"), "-output", out], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()
    assert filecmp.cmp(out, gold_ufi, shallow=False)
    # a missing input file is a loud error, exit status 1 (myutils.cpp:915)
    r = subprocess.run([exe, "-make_ufi", os.path.join(tmp_path, "nope.fa"), "-output", out], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 1


@pytest.mark.skipif(not ol.have_ref(), reason="reference binary oracle/_ref/urmap not built")
@pytest.mark.parametrize("nbases", [1200, 30011, 171717, 999983])
def test_cli_default_slot_count_equals_reference_binary(tmp_path, nbases):
    """GetPrime's ladder (prime.cpp:11-21) is regenerated, not stored: same default slot count as the reference binary
    for FASTA files of several sizes (also with -veryfast: MaxIx 3, ufindexio.cpp:133-136)."""
    import subprocess
    from urmap_amd import synth
    exe = os.path.join(ROOT, "urmap_amd", "urmap")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    d = str(tmp_path)
    g = synth.make_genome(nbases, [nbases - nbases // 3, nbases // 3], repeat_frac=0.2, n_families=3)
    synth.write_fasta(os.path.join(d, "x.fa"), g)
    for extra in ([], ["-veryfast"]):
        ol.run_ref(["-make_ufi", "x.fa", "-output", "ref.ufi"] + extra, cwd=d)
        r = subprocess.run([exe, "-make_ufi", os.path.join(d, "x.fa"), "-output", os.path.join(d, "mine.ufi")] + extra,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode == 0, r.stderr.decode()
        assert ol.ufi_header(os.path.join(d, "mine.ufi")) == ol.ufi_header(os.path.join(d, "ref.ufi"))
        assert filecmp.cmp(os.path.join(d, "mine.ufi"), os.path.join(d, "ref.ufi"), shallow=False)


def test_file_to_file_call_and_gpu_builder_fail_loudly_without_a_device(gold_ufi, tmp_path):
    """No GPU here: urmapx_map_files must come back with URMAPX_E_NODEVICE and a message (not exit, not fall back to any
    CPU mapping), urmapx_make_ufi_gpu likewise; the command line's -make_ufi says that it builds on the host and
    writes the reference's bytes."""
    import subprocess
    idx = api.Index.open(gold_ufi)
    with pytest.raises(api.UrmapxError) as e:
        api.map_files(idx, os.path.join(GOLD, "se150.fq"), samout=os.path.join(tmp_path, "x.sam"))
    assert e.value.code == api.E_NODEVICE and "Uploading index" in str(e.value)
    with pytest.raises(api.UrmapxError) as e2:
        api.make_ufi_gpu(0, os.path.join(GOLD, "g.fa"), os.path.join(tmp_path, "x.ufi"), 100003)
    assert e2.value.code == api.E_NODEVICE
    out = os.path.join(tmp_path, "cli.ufi")
    r = subprocess.run([os.path.join(ROOT, "urmap_amd", "urmap"), "-make_ufi", os.path.join(GOLD, "g.fa"), "-output", out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-500:]
    assert b"building on the host" in r.stderr
    assert filecmp.cmp(out, gold_ufi, shallow=False)
    r = subprocess.run([os.path.join(ROOT, "urmap_amd", "urmap"), "-map", os.path.join(GOLD, "se150.fq"), "-ufi", gold_ufi,
                        "-samout", os.path.join(tmp_path, "y.sam")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Fatal error" in r.stderr and b"Uploading index" in r.stderr


@pytest.mark.parametrize("key", ["load_factor_0.3", "load_factor_0.9_veryfast", "notrunclabels", "trunclabels_default"])
def test_cli_make_ufi_options_reproduce_reference_index(tmp_path, key):
    """cmd_make_ufi's options beyond -slots (ufindexio.cpp:117-150): -load_factor (slots = GetPrime(int64(size / LF))),
    -veryfast (MaxIx 3), -notrunclabels (whole label lines).  tests/golden/ufi_opts.json holds what the reference binary
    wrote for g.fa (make_golden.py ufiopts); the host builder (-host, no GPU needed) must write the same bytes."""
    import hashlib
    import json
    import subprocess
    case = json.load(open(os.path.join(GOLD, "ufi_opts.json")))[key]
    out = os.path.join(tmp_path, "o.ufi")
    log = os.path.join(tmp_path, "make.log")
    r = subprocess.run([os.path.join(ROOT, "urmap_amd", "urmap"), "-make_ufi", os.path.join(GOLD, "g.fa"), "-output", out, "-host",
                        "-log", log] + case["options"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    w, maxix, sds, slots = ol.ufi_header(out)
    assert (slots, maxix) == (case["slots"], case["max_ix"])
    assert [d[0] for d in ol.Index.load(out).directory()] == case["labels"]
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == case["sha256"]
    text = open(log).read()  # -log FILE: command line, start and finish stamps (myutils.cpp)
    assert "-make_ufi" in text and "Started " in text and "Finished " in text and "Elapsed time " in text
