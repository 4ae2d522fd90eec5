"""GPU tests of the text ends of `urmap -map` on the device (urmapx_text_map_se): a chunk of FASTQ bytes in, the bytes of
its SAM records out.  Checked against the reference's golden SAM files, against the oracle's SAM for seeded reads, and
against the host reader + host formatter of this library on the same bytes (FASTQSeqSource::GetNextLo,
fastqseqsource.cpp:9-116; State1::SetSAM, setsam.cpp:12-207)."""
import gzip
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _records(b):
    return [l for l in b.split(b"\n") if l and not l.startswith(b"@")]


@pytest.fixture(scope="module")
def golden(tmp_path_factory):
    from urmap_amd import api
    d = tmp_path_factory.mktemp("text")
    ufi = os.path.join(d, "g.ufi")
    with gzip.open(os.path.join(GOLD, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0)
    yield {"index": idx, "mapper": m}
    m.close()


@pytest.mark.parametrize("name", ["se150", "se250", "se_short"])
def test_golden_fastq_bytes_to_golden_sam_bytes(golden, name):
    from urmap_amd import api
    fq = open(os.path.join(GOLD, name + ".fq"), "rb").read()
    sam, rep = golden["mapper"].map_text_se(fq)
    assert rep["reason"] == api.TEXT_OK, rep
    want = _records(open(os.path.join(GOLD, name + ".sam"), "rb").read())
    assert _records(sam) == want
    assert sam.endswith(b"\n") and len(sam) == rep["sam_bytes"] and rep["records"] == len(want)
    mapped = [l for l in want if l.split(b"\t")[2] != b"*"]
    assert rep["unmapped"] == len(want) - len(mapped)
    assert rep["mapped_q"] == sum(1 for l in mapped if int(l.split(b"\t")[4]) >= 10)
    assert rep["mapped_q"] + rep["mapped_lowq"] + rep["unmapped"] == rep["records"]


def test_chunks_cut_at_record_ends_concatenate_to_the_whole(golden):
    fq = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    lines = fq.split(b"\n")[:-1]
    whole, _ = golden["mapper"].map_text_se(fq)
    parts = []
    for lo in range(0, len(lines), 4 * 37):
        chunk = b"\n".join(lines[lo:lo + 4 * 37]) + b"\n"
        sam, rep = golden["mapper"].map_text_se(chunk)
        assert rep["reason"] == 0 and rep["records"] == len(chunk.split(b"\n")) // 4
        parts.append(sam)
    assert b"".join(parts) == whole
    assert golden["mapper"].map_text_se(b"")[1]["records"] == 0


def test_copy_back_deferred_two_chunks_in_flight(golden):
    """urmapx_text_set_deferred: chunk i + 1 goes in before chunk i's text is waited for (what a lane of urmapx_map_files does).
    Chunks of different sizes (the device's SAM array grows under way), one rejected chunk in between, one whose buffer is too
    small (urmapx_text_fetch_sam, deferred too), an empty one: every chunk's bytes equal the one-at-a-time call's; a third
    chunk before a wait is refused."""
    import ctypes as C
    from urmap_amd import api
    m = golden["mapper"]
    fq = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    lines = fq.split(b"\n")[:-1]
    cuts = [0, 4 * 11, 4 * 12, 4 * 90, 4 * 91, 4 * 300, len(lines)]
    chunks = [b"\n".join(lines[a:b]) + b"\n" for a, b in zip(cuts, cuts[1:]) if b > a]
    chunks.insert(3, b"@r\r\nACGT\r\n+\r\nIIII\r\n")  # '\r': the device parser hands it back, no text is on its way for it
    chunks.insert(5, b"")
    want = [m.map_text_se(c) for c in chunks]
    caps = [None] * len(chunks)
    caps[2] = 4096  # far too small for 78 records
    got = m.map_text_se_stream(chunks, sam_caps=caps)
    assert len(got) == len(want)
    for i, ((gs, gr), (ws, wr)) in enumerate(zip(got, want)):
        assert gr["reason"] == wr["reason"], (i, gr, wr)
        if wr["reason"] == api.TEXT_OK:
            assert (gs or b"") == (ws or b""), i
            assert gr["records"] == wr["records"] and gr["sam_bytes"] == wr["sam_bytes"] and gr["mapped_q"] == wr["mapped_q"]
    assert got[3][1]["reason"] == api.TEXT_CR
    # twice in a row, and the plain call still works afterwards
    assert [g[0] for g in m.map_text_se_stream(chunks[:3])] == [w[0] for w in want[:3]]
    assert m.map_text_se(chunks[0])[0] == want[0][0]
    # a third chunk while two are in flight is refused; waiting makes room
    L = api.lib()
    assert L.urmapx_text_set_deferred(m._text, 1) == 0
    bufs = [np.empty(2 * len(chunks[0]) + 4096 * 64, np.uint8) for _ in range(3)]
    src = np.frombuffer(chunks[0], np.uint8)
    rep = api.TextReport()
    for k in range(2):
        assert L.urmapx_text_map_se(m._text, src.ctypes.data, len(src), 10, bufs[k].ctypes.data, len(bufs[k]), C.byref(rep)) == 0
        assert rep.reason == api.TEXT_DEFERRED
    assert L.urmapx_text_map_se(m._text, src.ctypes.data, len(src), 10, bufs[2].ctypes.data, len(bufs[2]), C.byref(rep)) == api.E_ARG
    assert L.urmapx_text_set_deferred(m._text, 0) == api.E_ARG  # not while texts are on their way
    for k in range(2):
        assert L.urmapx_text_wait(m._text, C.byref(rep)) == 0 and rep.reason == api.TEXT_OK
        assert bufs[k][: rep.sam_bytes].tobytes() == want[0][0]
    assert L.urmapx_text_wait(m._text, C.byref(rep)) == api.E_ARG  # nothing left to wait for
    assert L.urmapx_text_set_deferred(m._text, 0) == 0


def _fastq_text(reads, labels=None):
    out = []
    for i, (lab, s, q) in enumerate(reads):
        lab = labels[i] if labels else lab
        out.append(b"@" + (lab if isinstance(lab, bytes) else lab.encode()) + b"\n" + bytes(s) + b"\n+\n" + bytes(q) + b"\n")
    return b"".join(out)


@pytest.mark.parametrize("read_len,sub,indel,n", [(150, 0.03, 0.01, 3000), (250, 0.04, 0.02, 1500), (64, 0.02, 0.0, 1000),
                                                   (700, 0.03, 0.01, 300)])
def test_seeded_reads_equal_oracle_sam_and_host_formatter(small_case, tmp_path, read_len, sub, indel, n):
    """Gapped alignments (CIGAR from the path, dangling-M rule), both strands, labels with blanks and /1 /2 endings, lower
    case and IUPAC letters: the device text equals the oracle's SAM file and this library's host formatter."""
    from urmap_amd import api, synth
    from test_gpu_parity import mutate_edge_reads
    idx = api.Index.open(small_case["ufi"]).upload(0)
    m = api.Mapper(idx, device=0)
    reads = synth.make_reads(4000 + read_len, small_case["genome"], n, read_len=read_len, sub=sub, ins=indel / 2, dele=indel / 2,
                             random_frac=0.05)
    reads = mutate_edge_reads(reads, read_len)
    labels = []
    for i, (lab, _, _) in enumerate(reads):
        lab = lab if isinstance(lab, str) else lab.decode()
        if i % 5 == 1: lab += " extra words\there"
        if i % 5 == 2: lab += "/1"
        if i % 5 == 3: lab += "/2 tail"
        if i % 5 == 4: lab = lab + "/3"
        if i % 97 == 0: lab = "x" * 150 + lab
        if i % 101 == 0: lab = "\tleading blank"
        labels.append(lab)
    fq_text = _fastq_text(reads, labels)
    fq = os.path.join(tmp_path, "r.fq")
    open(fq, "wb").write(fq_text)
    sam, rep = m.map_text_se(fq_text)
    assert rep["reason"] == api.TEXT_OK and rep["records"] == n
    osam = os.path.join(tmp_path, "o.sam")
    small_case["oracle_index"].map_file_se(fq, osam, threads=4)
    assert _records(sam) == _records(open(osam, "rb").read())
    # the host reader and formatter of the library on the same bytes
    lab2, bases, offs, quals = api.read_fastq_arrays(fq)
    res, ops = m.map_se(bases, offs)
    host = idx.sam_se(res, ops, lab2, bases, offs, quals)
    assert sam == host
    gapped = sum(1 for l in _records(sam) if (b"I" in l.split(b"\t")[5] or b"D" in l.split(b"\t")[5]))
    assert indel == 0 or gapped > n // 50
    m.close()
    idx.close()


def test_chunks_the_device_parser_hands_back(golden):
    """'\\r', a missing final newline, a line count that is not a multiple of four, a malformed record, a blank line: nothing is
    written and the reason says why (the host reader then deals with the chunk the way the reference does)."""
    from urmap_amd import api
    m = golden["mapper"]
    fq = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    lines = fq.split(b"\n")[:-1]
    ok = b"\n".join(lines[:40]) + b"\n"
    assert m.map_text_se(ok)[1]["reason"] == api.TEXT_OK
    assert m.map_text_se(ok.replace(b"\n", b"\r\n"))[1]["reason"] == api.TEXT_CR
    assert m.map_text_se(ok[:-1])[1]["reason"] == api.TEXT_RAGGED
    assert m.map_text_se(b"\n".join(lines[:39]) + b"\n")[1]["reason"] == api.TEXT_RAGGED
    bad = list(lines[:40])
    bad[5] = bad[5][:-1] + b"1"  # a digit among the bases
    assert m.map_text_se(b"\n".join(bad) + b"\n")[1]["reason"] == api.TEXT_BAD_RECORD
    bad = list(lines[:40])
    bad[7] = bad[7][:-1]  # one quality byte short
    assert m.map_text_se(b"\n".join(bad) + b"\n")[1]["reason"] == api.TEXT_BAD_RECORD
    bad = list(lines[:40])
    bad[8] = b"r" + bad[8][1:]  # '@' missing
    assert m.map_text_se(b"\n".join(bad) + b"\n")[1]["reason"] == api.TEXT_BAD_RECORD
    bad = list(lines[:36]) + [b"", b"", b"", b""]
    assert m.map_text_se(b"\n".join(bad) + b"\n")[1]["reason"] == api.TEXT_BAD_RECORD
    assert m.map_text_se(b"\n" * 1000)[1]["reason"] in (api.TEXT_RAGGED, api.TEXT_BAD_RECORD)
    whole = m.map_text_se(ok)[0]
    sam, rep = m.map_text_se(ok, sam_cap=100)
    assert sam is None and rep["reason"] == api.TEXT_SAM_CAP and rep["sam_bytes"] == len(whole)
    # the chunk is mapped; its text is fetched into a buffer that is large enough, without another search
    assert m.fetch_text_sam(200)[1]["reason"] == api.TEXT_SAM_CAP
    sam, rep = m.fetch_text_sam(len(whole))
    assert sam == whole and rep["reason"] == api.TEXT_OK and rep["records"] == 10
    with pytest.raises(api.UrmapxError):
        m.fetch_text_sam(len(whole))


def test_reads_outside_the_device_domain_are_counted(golden, tmp_path):
    """A read longer than URMAPX_MAX_QL, one shorter than the word and an empty one: their records are the host
    formatter's for the same flagged results, and the report counts them."""
    from urmap_amd import api
    m, idx = golden["mapper"], golden["index"]
    rng = np.random.default_rng(5)
    long_read = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 1500))
    fq_text = b"@long\n" + long_read + b"\n+\n" + b"I" * 1500 + b"\n@short\nACGT\n+\nIIII\n@empty\n\n+\n\n"
    sam, rep = m.map_text_se(fq_text)
    assert rep["reason"] == api.TEXT_OK and rep["records"] == 3 and rep["unsupported"] >= 1
    fq = os.path.join(tmp_path, "odd.fq")
    open(fq, "wb").write(fq_text)
    lab, bases, offs, quals = api.read_fastq_arrays(fq)
    res, ops = m.map_se(bases, offs, allow_unsupported=True)
    assert sam == idx.sam_se(res, ops, lab, bases, offs, quals)
    assert rep["unsupported"] == int((res["status"] != 0).sum())
    recs = sam.split(b"\n")
    assert recs[0].split(b"\t")[0] == b"long" and recs[0].split(b"\t")[9] in (long_read,)
    assert recs[1] == b"short\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII"


# ---- the command line: text phase, hand-back to the host reader in the middle of a file ----
EXE = os.path.join(ROOT, "urmap_amd", "urmap")


def _run_cli(fq, ufi, out, batch, host_text=False, extra=()):
    import subprocess
    env = dict(os.environ)
    env.pop("URMAPX_HOST_TEXT", None)
    if host_text:
        env["URMAPX_HOST_TEXT"] = "1"
    return subprocess.run([EXE, "-map", fq, "-ufi", ufi, "-samout", out, "-batch", str(batch), *extra], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300, env=env)


@pytest.fixture(scope="module")
def golden_ufi(tmp_path_factory):
    d = tmp_path_factory.mktemp("textcli")
    ufi = os.path.join(d, "g.ufi")
    with gzip.open(os.path.join(GOLD, "g.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    return ufi


def _sam_body(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")]


@pytest.mark.parametrize("batch", [7, 64, 1000])
@pytest.mark.parametrize("variant", ["plain", "crlf_second_half", "cr_in_one_record", "no_final_newline", "blank_lines_at_end",
                                     "long_labels"])
def test_cli_text_phase_and_host_reader_write_the_same_file(golden_ufi, tmp_path, variant, batch):
    """Files the device parser takes whole, and files where it hands a chunk back in the middle ('\\r', a ragged end): the SAM
    file is the reference's golden SAM either way, and equal byte for byte to what the host-reader-only pipeline writes."""
    fq = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    lines = fq.split(b"\n")[:-1]
    if variant == "crlf_second_half":
        text = b"\n".join(lines[:800]) + b"\n" + b"\r\n".join(lines[800:]) + b"\r\n"
    elif variant == "cr_in_one_record":
        text = b"\n".join(lines[:1001]) + b"\r\n" + b"\n".join(lines[1001:]) + b"\n"
    elif variant == "no_final_newline":
        text = fq[:-1]
    elif variant == "blank_lines_at_end":
        text = fq + b"\n\n\n"
    elif variant == "long_labels":
        text = b"".join((l + b" " + b"z" * (i % 300) if i % 4 == 0 else l) + b"\n" for i, l in enumerate(lines))
    else:
        text = fq
    src = os.path.join(tmp_path, "in.fq")
    open(src, "wb").write(text)
    a, b = os.path.join(tmp_path, "text.sam"), os.path.join(tmp_path, "host.sam")
    r1 = _run_cli(src, golden_ufi, a, batch)
    r2 = _run_cli(src, golden_ufi, b, batch, host_text=True)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr.decode()[-1500:], r2.stderr.decode()[-1500:])
    assert _sam_body(a) == _sam_body(b)
    assert _sam_body(a) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]
    # the HitStats report (stderr) is the same too
    tail = lambda r: [l for l in r.stderr.decode().splitlines() if ("%" in l or "reads" in l.lower()) and "/sec" not in l]
    assert tail(r1) == tail(r2)


@pytest.mark.parametrize("batch", [5, 50])
@pytest.mark.parametrize("damage,needle", [("digit_in_bases", "Invalid sequence letter"), ("short_qual", "Bad FASTQ record"),
                                           ("no_at", "expected '@'"), ("blank_in_middle", "Empty line nr"),
                                           ("truncated", "Unexpected end-of-file")])
def test_cli_malformed_record_late_in_the_file_dies_with_the_reference_message(golden_ufi, tmp_path, damage, needle, batch):
    """The record is in a later chunk: the device parser hands that chunk back, the host reader takes over at its first byte
    with the line count kept, and the message (with its line number) is the one the host-only pipeline gives."""
    fq = open(os.path.join(GOLD, "se150.fq"), "rb").read()
    lines = fq.split(b"\n")[:-1]
    k = 4 * 301  # first line of record 301
    if damage == "digit_in_bases":
        lines[k + 1] = lines[k + 1][:20] + b"7" + lines[k + 1][21:]
    elif damage == "short_qual":
        lines[k + 3] = lines[k + 3][:-3]
    elif damage == "no_at":
        lines[k] = b"r" + lines[k][1:]
    elif damage == "blank_in_middle":
        lines.insert(k, b"")
    elif damage == "truncated":
        lines = lines[: k + 2]
    src = os.path.join(tmp_path, "bad.fq")
    open(src, "wb").write(b"\n".join(lines) + b"\n")
    r1 = _run_cli(src, golden_ufi, os.path.join(tmp_path, "a.sam"), batch)
    r2 = _run_cli(src, golden_ufi, os.path.join(tmp_path, "b.sam"), batch, host_text=True)
    assert r1.returncode == 1 and r2.returncode == 1
    m1 = [l for l in r1.stderr.decode().splitlines() if needle in l]
    m2 = [l for l in r2.stderr.decode().splitlines() if needle in l]
    assert m1 and m1 == m2, (r1.stderr.decode()[-800:], r2.stderr.decode()[-800:])


def test_map_files_reports_the_same_counters_with_and_without_the_text_phase(golden, tmp_path):
    from urmap_amd import api
    fq = os.path.join(GOLD, "se250.fq")
    a, b = os.path.join(tmp_path, "a.sam"), os.path.join(tmp_path, "b.sam")
    os.environ.pop("URMAPX_HOST_TEXT", None)
    r1 = api.map_files(golden["index"], fq, samout=a, batch=33, streams=2, cmdline="t")
    os.environ["URMAPX_HOST_TEXT"] = "1"
    try:
        r2 = api.map_files(golden["index"], fq, samout=b, batch=33, streams=2, cmdline="t")
    finally:
        os.environ.pop("URMAPX_HOST_TEXT", None)
    for k in ("reads", "mapped_q", "mapped_lowq", "unmapped", "unsupported"):
        assert r1[k] == r2[k], k
    assert open(a, "rb").read() == open(b, "rb").read()
    api.lib().urmapx_host_pool_trim()


def test_lanes_are_kept_between_calls_and_go_with_their_index(small_case, tmp_path, monkeypatch):
    """urmapx_map_files keeps its lanes' mapping contexts (LanePool, pipeline.cpp): the second call on an index allocates no device
    array (report.alloc_dev_calls == 0), a pair run after single-end runs reuses the same contexts, the SAM is the oracle's every
    time; urmapx_host_pool_trim and Index.close destroy the kept lanes (the next call allocates again), and so does
    URMAPX_NO_LANE_POOL=1 for a call."""
    from urmap_amd import api, synth
    for k in list(os.environ):
        if k.startswith("URMAPX_TEST_") or k.startswith("URMAPX_DEBUG_"):
            monkeypatch.delenv(k)  # (such knobs switch the pool off: contexts cache what they decide)
    monkeypatch.delenv("URMAPX_NO_LANE_POOL", raising=False)
    oi = small_case["oracle_index"]
    fq, fq2 = os.path.join(tmp_path, "a.fq"), os.path.join(tmp_path, "b.fq")
    synth.write_fastq(fq, synth.make_reads(31, small_case["genome"], 6000, read_len=150, sub=0.01, ins=0.001, dele=0.001))
    synth.write_fastq(fq2, synth.make_reads(32, small_case["genome"], 7000, read_len=150, sub=0.02, ins=0.002, dele=0.002))
    r1, r2 = synth.make_pairs(33, small_case["genome"], 3000, read_len=150, sub1=0.01, sub2=0.02, ins=0.001, dele=0.001)
    m1, m2 = os.path.join(tmp_path, "m1.fq"), os.path.join(tmp_path, "m2.fq")
    synth.write_fastq(m1, r1)
    synth.write_fastq(m2, r2)
    want, want2, wantp = (os.path.join(tmp_path, n) for n in ("o1.sam", "o2.sam", "op.sam"))
    oi.map_file_se(fq, want, threads=4)
    oi.map_file_se(fq2, want2, threads=4)
    oi.map_file_pe(m1, m2, wantp, threads=4)
    body = lambda p: _records(open(p, "rb").read())
    api.lib().urmapx_host_pool_trim()
    idx = api.Index.open(small_case["ufi"]).upload(0)
    out = os.path.join(tmp_path, "x.sam")
    a = api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")
    assert a["alloc_dev_calls"] > 0 and body(out) == body(want)
    b = api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")
    assert b["alloc_dev_calls"] == 0 and b["alloc_pinned_calls"] == 0 and body(out) == body(want)
    c = api.map_files(idx, fq2, samout=out, batch=1000, streams=2, cmdline="t")  # another file in smaller chunks: nothing grows
    assert c["alloc_dev_calls"] == 0 and body(out) == body(want2)
    p = api.map_files(idx, m1, m2, samout=out, batch=1000, streams=2, cmdline="t")  # the pair path on the contexts the single-end runs left
    assert body(out) == body(wantp)
    p2 = api.map_files(idx, m1, m2, samout=out, batch=1000, streams=2, cmdline="t")
    assert p2["alloc_dev_calls"] == 0 and body(out) == body(wantp)
    monkeypatch.setenv("URMAPX_NO_LANE_POOL", "1")
    d = api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")
    assert d["alloc_dev_calls"] > 0 and body(out) == body(want)
    monkeypatch.delenv("URMAPX_NO_LANE_POOL")
    assert api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")["alloc_dev_calls"] == 0  # (the kept lanes were not touched)
    api.lib().urmapx_host_pool_trim()
    e = api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")
    assert e["alloc_dev_calls"] > 0 and e["alloc_pinned_calls"] > 0 and body(out) == body(want)
    idx.close()  # destroys the lanes kept for it
    idx = api.Index.open(small_case["ufi"]).upload(0)
    f = api.map_files(idx, fq, samout=out, batch=1500, streams=2, cmdline="t")
    assert f["alloc_dev_calls"] > 0 and body(out) == body(want)
    idx.close()
    api.lib().urmapx_host_pool_trim()


# ---- pairs ----
@pytest.mark.parametrize("name,ufi_gz", [("pe150", "g.ufi.gz"), ("pe100_noisy", "g.ufi.gz"), ("pe120_rep", "r.ufi.gz")])
def test_golden_mate_files_to_golden_pair_sam(tmp_path, name, ufi_gz):
    """urmapx_text_map_pe: the bytes of the two mate files -> the reference's golden -map2 SAM (flags, RNEXT, PNEXT, TLEN
    of SetSAM2, output2.cpp:61-128)."""
    from urmap_amd import api
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(GOLD, ufi_gz), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0)
    f1 = open(os.path.join(GOLD, name + "_1.fq"), "rb").read()
    f2 = open(os.path.join(GOLD, name + "_2.fq"), "rb").read()
    sam, rep = m.map_text_pe(f1, f2)
    assert rep["reason"] == api.TEXT_OK, rep
    want = _records(open(os.path.join(GOLD, name + ".sam"), "rb").read())
    assert _records(sam) == want
    assert rep["records"] == len(want) and rep["mapped_q"] + rep["mapped_lowq"] + rep["unmapped"] == len(want)
    # chunks with different record counts, or a damaged mate file, are handed back
    l2 = f2.split(b"\n")[:-1]
    assert m.map_text_pe(f1, b"\n".join(l2[:-4]) + b"\n")[1]["reason"] == api.TEXT_UNEQUAL
    assert m.map_text_pe(f1, f2.replace(b"\n", b"\r\n", 3))[1]["reason"] == api.TEXT_CR
    assert m.map_text_pe(f1, b"\n".join(l2[:-1]) + b"\n")[1]["reason"] == api.TEXT_RAGGED
    bad = list(l2)
    bad[9] = bad[9][:-2]
    assert m.map_text_pe(f1, b"\n".join(bad) + b"\n")[1]["reason"] == api.TEXT_BAD_RECORD
    m.close()
    idx.close()


def _run_cli2(fq1, fq2, ufi, out, batch, host_text=False, extra=()):
    import subprocess
    env = dict(os.environ)
    env.pop("URMAPX_HOST_TEXT", None)
    if host_text:
        env["URMAPX_HOST_TEXT"] = "1"
    return subprocess.run([EXE, "-map2", fq1, "-reverse", fq2, "-ufi", ufi, "-samout", out, "-batch", str(batch), *extra],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)


@pytest.mark.parametrize("batch", [6, 50, 2000])
@pytest.mark.parametrize("variant", ["plain", "crlf_in_mates_from_the_middle", "no_final_newline_in_one", "longer_labels_in_mates"])
def test_cli_map2_text_phase_and_host_reader_write_the_same_file(golden_ufi, tmp_path, variant, batch):
    f1 = open(os.path.join(GOLD, "pe150_1.fq"), "rb").read()
    f2 = open(os.path.join(GOLD, "pe150_2.fq"), "rb").read()
    l2 = f2.split(b"\n")[:-1]
    if variant == "crlf_in_mates_from_the_middle":
        f2 = b"\n".join(l2[:600]) + b"\n" + b"\r\n".join(l2[600:]) + b"\r\n"
    elif variant == "no_final_newline_in_one":
        f2 = f2[:-1]
    elif variant == "longer_labels_in_mates":
        f2 = b"".join((l + b" " + b"q" * (i % 211) if i % 4 == 0 else l) + b"\n" for i, l in enumerate(l2))
    a1, a2 = os.path.join(tmp_path, "m1.fq"), os.path.join(tmp_path, "m2.fq")
    open(a1, "wb").write(f1)
    open(a2, "wb").write(f2)
    a, b = os.path.join(tmp_path, "text.sam"), os.path.join(tmp_path, "host.sam")
    r1 = _run_cli2(a1, a2, golden_ufi, a, batch)
    r2 = _run_cli2(a1, a2, golden_ufi, b, batch, host_text=True)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr.decode()[-1500:], r2.stderr.decode()[-1500:])
    assert _sam_body(a) == _sam_body(b)
    if variant != "longer_labels_in_mates":  # text after "/2" keeps the "/2" in QNAME (setsam.cpp:36-38), in both pipelines
        assert _sam_body(a) == [l for l in open(os.path.join(GOLD, "pe150.sam"), "rb").read().split(b"\n") if l]


@pytest.mark.parametrize("batch", [10, 100])
@pytest.mark.parametrize("which,cut", [(1, 4 * 120), (2, 4 * 120), (2, 4 * 120 + 2)])
def test_cli_map2_unequal_mate_files_die_with_the_host_pipelines_message(golden_ufi, tmp_path, which, cut, batch):
    f1 = open(os.path.join(GOLD, "pe150_1.fq"), "rb").read().split(b"\n")[:-1]
    f2 = open(os.path.join(GOLD, "pe150_2.fq"), "rb").read().split(b"\n")[:-1]
    if which == 1:
        f1 = f1[:cut]
    else:
        f2 = f2[:cut]
    a1, a2 = os.path.join(tmp_path, "m1.fq"), os.path.join(tmp_path, "m2.fq")
    open(a1, "wb").write(b"\n".join(f1) + b"\n")
    open(a2, "wb").write(b"\n".join(f2) + b"\n")
    r1 = _run_cli2(a1, a2, golden_ufi, os.path.join(tmp_path, "a.sam"), batch)
    r2 = _run_cli2(a1, a2, golden_ufi, os.path.join(tmp_path, "b.sam"), batch, host_text=True)
    assert r1.returncode == 1 and r2.returncode == 1
    last = lambda r: [l for l in r.stderr.decode().splitlines() if l.strip()][-1]
    assert last(r1) == last(r2), (r1.stderr.decode()[-600:], r2.stderr.decode()[-600:])


def test_short_reads_outgrow_the_sam_buffer_and_are_fetched_again(small_case, tmp_path):
    """32-base reads: the SAM text is 1.5 x the FASTQ text, more than the lane's buffer for the chunk (sized for 1.12 x +
    1 MB), so the chunk's text is fetched a second time into a larger buffer without another search
    (urmapx_text_fetch_sam).  Same bytes as the host-only pipeline."""
    from urmap_amd import synth
    reads = synth.make_reads(77, small_case["genome"], 150_000, read_len=32, sub=0.01, ins=0.0, dele=0.0, random_frac=0.05)
    fq = os.path.join(tmp_path, "short.fq")
    synth.write_fastq(fq, reads)
    assert os.path.getsize(fq) > 8_000_000
    a, b = os.path.join(tmp_path, "text.sam"), os.path.join(tmp_path, "host.sam")
    r1 = _run_cli(fq, small_case["ufi"], a, 1 << 20)
    r2 = _run_cli(fq, small_case["ufi"], b, 1 << 20, host_text=True)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr.decode()[-800:], r2.stderr.decode()[-800:])
    assert os.path.getsize(a) > 1.12 * 1.04 * os.path.getsize(fq) + (1 << 20)  # more than the first buffer holds
    assert _sam_body(a) == _sam_body(b)


def test_target_label_too_long_for_the_device_formatter_goes_to_the_host(tmp_path):
    """A sequence label of 300 bytes: the device formatter hands the chunk back (URMAPX_TEXT_LONG_NAME), the pipeline
    formats on the host, and the SAM is the oracle's."""
    import oracle_lib as ol
    from urmap_amd import api, synth
    g = synth.make_genome(9, [50000, 20000], repeat_frac=0.2, n_families=4)
    g = [("L" * 300, g[0][1]), g[1]]
    fa = os.path.join(tmp_path, "g.fa")
    synth.write_fasta(fa, g)
    oi = ol.Index.build(fa, 131101)
    ufi = os.path.join(tmp_path, "g.ufi")
    oi.save(ufi)
    reads = synth.make_reads(10, g, 500, read_len=100, sub=0.01, ins=0.0, dele=0.0)
    fq = os.path.join(tmp_path, "r.fq")
    synth.write_fastq(fq, reads)
    idx = api.Index.open(ufi).upload(0)
    m = api.Mapper(idx, device=0)
    sam, rep = m.map_text_se(open(fq, "rb").read())
    assert sam is None and rep["reason"] == api.TEXT_LONG_NAME
    m.close()
    idx.close()
    a, osam = os.path.join(tmp_path, "text.sam"), os.path.join(tmp_path, "o.sam")
    r1 = _run_cli(fq, ufi, a, 100)
    assert r1.returncode == 0, r1.stderr.decode()[-800:]
    oi.map_file_se(fq, osam, threads=2)
    assert _records(open(a, "rb").read()) == _records(open(osam, "rb").read())
    assert any(b"L" * 300 in l for l in _records(open(a, "rb").read()))


@pytest.mark.parametrize("batch", [20000, 80000])
def test_cli_map2_mate_file_with_much_longer_records(small_case, tmp_path, batch):
    """The second mate file's records are five times as long as the first's (a long comment after each label): the mate
    chunk outgrows the buffer sized from the first file and the reader moves to a larger one, mid-file (batch 20000) and
    on the last chunk (batch 80000).  Same SAM as the host-only pipeline."""
    from urmap_amd import synth
    r1, r2 = synth.make_pairs(5, small_case["genome"], 20000, read_len=100)
    f1, f2 = os.path.join(tmp_path, "m1.fq"), os.path.join(tmp_path, "m2.fq")
    synth.write_fastq(f1, r1)
    with open(f2, "wb") as f:
        for lab, s, q in r2:
            lab = lab if isinstance(lab, bytes) else lab.encode()
            f.write(b"@" + lab + b" " + b"c" * 1000 + b"\n" + bytes(s) + b"\n+\n" + bytes(q) + b"\n")
    assert os.path.getsize(f2) > 4 * os.path.getsize(f1)
    a, b = os.path.join(tmp_path, "text.sam"), os.path.join(tmp_path, "host.sam")
    ra = _run_cli2(f1, f2, small_case["ufi"], a, batch)
    rb = _run_cli2(f1, f2, small_case["ufi"], b, batch, host_text=True)
    assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr.decode()[-800:], rb.stderr.decode()[-800:])
    assert _sam_body(a) == _sam_body(b)
    assert len(_sam_body(a)) > 40000


@pytest.mark.parametrize("host_text", [False, True])
def test_cli_samout_to_a_pipe(golden_ufi, tmp_path, host_text):
    """`-samout /dev/stdout | cat`: the SAM file is a pipe; records go out in order with write() (the reference writes its
    SAM through stdio, so pipes work there: outfiles.cpp:7-12)."""
    import subprocess
    env = dict(os.environ)
    env.pop("URMAPX_HOST_TEXT", None)
    if host_text:
        env["URMAPX_HOST_TEXT"] = "1"
    out = os.path.join(tmp_path, "piped.sam")
    cmd = f"'{EXE}' -map '{os.path.join(GOLD, 'se150.fq')}' -ufi '{golden_ufi}' -samout /dev/stdout -batch 50 2> '{tmp_path}/err.txt' | cat > '{out}'"
    r = subprocess.run(["bash", "-c", "set -o pipefail; " + cmd], timeout=300, env=env)
    assert r.returncode == 0, open(os.path.join(tmp_path, "err.txt")).read()[-800:]
    assert _sam_body(out) == [l for l in open(os.path.join(GOLD, "se150.sam"), "rb").read().split(b"\n") if l]


@pytest.mark.gpu
@pytest.mark.parametrize("paired", [False, True])
def test_cli_veryfast_through_the_device_text_path(small_case, tmp_path, paired):
    """`urmap -map ... -veryfast` (State1 method 7, state1.cpp:166-179) and `-map2 ... -veryfast` (State2::Search5, band
    radius 4, map2.cpp:17-21,47-49) with FASTQ bytes parsed and SAM bytes written on the device (pipeline.cpp switches the
    lanes' parameters), against the oracle's SAM; the host text stages must give the same file."""
    import subprocess
    import oracle_lib as ol
    from urmap_amd import synth
    exe = os.path.join(ROOT, "urmap_amd", "urmap")
    osam, sam = os.path.join(tmp_path, "o.sam"), os.path.join(tmp_path, "g.sam")
    if paired:
        r1, r2 = synth.make_pairs(515, small_case["genome"], 1500, read_len=150, sub1=0.02, sub2=0.04, ins=0.003, dele=0.003)
        f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
        synth.write_fastq(f1, r1)
        synth.write_fastq(f2, r2)
        small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4, veryfast=True)
        args = ["-map2", f1, "-reverse", f2, "-ufi", small_case["ufi"]]
    else:
        ufi = os.path.join(tmp_path, "vf.ufi")
        oi = ol.Index.build(small_case["fasta"], 524309, max_ix=3)  # the index -make_ufi -veryfast writes (MaxIx 3)
        oi.save(ufi)
        fq = os.path.join(tmp_path, "r.fq")
        synth.write_fastq(fq, synth.make_reads(516, small_case["genome"], 3000, read_len=150, sub=0.02, ins=0.002, dele=0.002, random_frac=0.03))
        oi.map_file_se(fq, osam, method=7, threads=4)
        args = ["-map", fq, "-ufi", ufi]
    want = [l for l in open(osam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
    assert sum(1 for l in want if l.split(b"\t")[2] != b"*") > len(want) * 0.8
    for env in ({}, {"URMAPX_HOST_TEXT": "1"}):
        r = subprocess.run([exe] + args + ["-samout", sam, "-veryfast", "-batch", "512"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env={**os.environ, **env})
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        got = [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@")]
        assert got == want, env


def _bgzf(data: bytes, block=60000) -> bytes:
    """BGZF (the bgzip / htslib container): gzip members of <= 64 KB, each with a 'BC' extra field holding its size - 1."""
    import struct
    import zlib
    out = bytearray()
    for i in list(range(0, len(data), block)) + [len(data)]:  # the last, empty block is BGZF's end-of-file marker
        raw = data[i:i + block] if i < len(data) else b""
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        bsize = 18 + len(comp) + 8
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
        out += comp + struct.pack("<II", zlib.crc32(raw) & 0xFFFFFFFF, len(raw))
    return bytes(out)


def _urmap(args, env=None):
    import subprocess
    r = subprocess.run([os.path.join(ROOT, "urmap_amd", "urmap")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env={**os.environ, "URMAPX_VERBOSE": "1", **(env or {})})
    return r.returncode, r.stderr.decode()


def _file_records(path):
    return [l for l in open(path, "rb").read().split(b"\n") if l and not l.startswith(b"@")]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gzip", "members", "bgzf", "padded"])
def test_cli_gz_input_takes_the_device_text_path(small_case, tmp_path, kind):
    """.gz FASTQ (linereader.cpp:14-113 reads it through zlib): chunks are cut out of the inflated stream and go to the device
    as bytes -- no host parsing, no host formatting (format 0.00 in the stage report), the oracle's records, at several chunk
    sizes; one gzip member, several members back to back, and a BGZF file (blocks inflated in parallel)."""
    from urmap_amd import synth
    reads = synth.make_reads(811, small_case["genome"], 6000, read_len=150, sub=0.02, ins=0.002, dele=0.002, random_frac=0.02)
    fq, osam, sam = (os.path.join(tmp_path, n) for n in ("r.fq", "o.sam", "g.sam"))
    synth.write_fastq(fq, reads)
    small_case["oracle_index"].map_file_se(fq, osam, threads=4)
    data = open(fq, "rb").read()
    gz = os.path.join(tmp_path, "r.fq.gz")
    if kind == "gzip":
        open(gz, "wb").write(gzip.compress(data, 1))
    elif kind == "members":
        cuts = [0, len(data) // 3 + 17, 2 * len(data) // 3 + 5, len(data)]  # member ends fall inside records
        open(gz, "wb").write(b"".join(gzip.compress(data[a:b], 1) for a, b in zip(cuts, cuts[1:])))
    elif kind == "padded":  # zero padding behind the last member: zlib's gzread (the reference, the host reader) ignores it
        open(gz, "wb").write(gzip.compress(data[: len(data) // 2], 1) + gzip.compress(data[len(data) // 2:], 1) + b"\0" * 1024)
    else:
        open(gz, "wb").write(_bgzf(data))
    want = _file_records(osam)
    for batch in ("100000", "700", "64"):
        rc, err = _urmap(["-map", gz, "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch])
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want, batch
        assert "format 0.00" in err, err[-600:]  # the SAM text was written on the device


@pytest.mark.gpu
def test_cli_bgzf_block_with_a_wrong_crc_is_refused(small_case, tmp_path):
    """The reference reads BGZF through zlib's gzread, which checks every member's CRC-32; the parallel block reader inflates raw
    deflate data and checks the trailer's CRC itself (round 5): a block whose stored CRC is off by a bit ends the run with an error,
    as a damaged plain gzip file does; the intact file maps."""
    from urmap_amd import synth
    reads = synth.make_reads(812, small_case["genome"], 3000, read_len=150, sub=0.02, ins=0.002, dele=0.002)
    fq, sam = os.path.join(tmp_path, "r.fq"), os.path.join(tmp_path, "g.sam")
    synth.write_fastq(fq, reads)
    good = _bgzf(open(fq, "rb").read())
    gz = os.path.join(tmp_path, "r.fq.gz")
    open(gz, "wb").write(good)
    rc, err = _urmap(["-map", gz, "-ufi", small_case["ufi"], "-samout", sam])
    assert rc == 0, err[-1000:]
    n_ok = len(_file_records(sam))
    assert n_ok == 3000
    first = (good[16] | (good[17] << 8)) + 1  # size of the first block
    bad = bytearray(good)
    bad[first - 8] ^= 0x01  # lowest byte of the first block's CRC-32
    open(gz, "wb").write(bytes(bad))
    rc, err = _urmap(["-map", gz, "-ufi", small_case["ufi"], "-samout", sam])
    assert rc != 0, err[-500:]
    plain_bad = bytearray(gzip.compress(open(fq, "rb").read(), 1))
    plain_bad[-8] ^= 0x01
    open(gz, "wb").write(bytes(plain_bad))
    rc2, err2 = _urmap(["-map", gz, "-ufi", small_case["ufi"], "-samout", sam])
    assert rc2 != 0, err2[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("second", ["gz", "plain", "bgzf"])
def test_cli_gz_pairs_take_the_device_text_path(small_case, tmp_path, second):
    from urmap_amd import synth
    r1, r2 = synth.make_pairs(812, small_case["genome"], 3000, read_len=150, sub1=0.02, sub2=0.03, ins=0.002, dele=0.002)
    r2 = [(lab + " some longer label text", s, q) if k % 5 == 0 else (lab, s, q) for k, (lab, s, q) in enumerate(r2)]  # records of unequal size in the two files
    f1, f2, osam, sam = (os.path.join(tmp_path, n) for n in ("r1.fq", "r2.fq", "o.sam", "g.sam"))
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4)
    g1 = f1 + ".gz"
    open(g1, "wb").write(gzip.compress(open(f1, "rb").read(), 1))
    g2 = f2
    if second == "gz":
        g2 = f2 + ".gz"
        open(g2, "wb").write(gzip.compress(open(f2, "rb").read(), 1))
    elif second == "bgzf":
        g2 = f2 + ".gz"
        open(g2, "wb").write(_bgzf(open(f2, "rb").read()))
    want = _file_records(osam)
    for batch in ("100000", "512"):
        rc, err = _urmap(["-map2", g1, "-reverse", g2, "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch])
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want, batch
        assert "format 0.00" in err, err[-600:]


@pytest.mark.gpu
def test_cli_gz_hand_back_and_errors(small_case, tmp_path):
    """What the device parser does not take -- '\\r' from the middle of the stream on, a damaged record -- goes back to the
    host reader, which finds its place in the .gz by uncompressed offset (gzseek): same SAM, same message and line number as
    with the host text stages; a truncated .gz is an error."""
    from urmap_amd import synth
    reads = synth.make_reads(813, small_case["genome"], 4000, read_len=150, sub=0.02)
    fq, osam, sam, sam2 = (os.path.join(tmp_path, n) for n in ("r.fq", "o.sam", "g.sam", "h.sam"))
    synth.write_fastq(fq, reads)
    small_case["oracle_index"].map_file_se(fq, osam, threads=4)
    want = _file_records(osam)
    data = open(fq, "rb").read()
    half = data.index(b"\n@", len(data) // 2) + 1
    gz = os.path.join(tmp_path, "crlf.fq.gz")
    open(gz, "wb").write(gzip.compress(data[:half] + data[half:].replace(b"\n", b"\r\n"), 1))
    for batch in ("100000", "300"):
        rc, err = _urmap(["-map", gz, "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch])
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want
    lines = data.split(b"\n")
    lines[4 * 3000 + 3] = lines[4 * 3000 + 3][:-5]  # a quality line shorter than its bases, late in the file
    bad = os.path.join(tmp_path, "bad.fq.gz")
    open(bad, "wb").write(gzip.compress(b"\n".join(lines), 1))
    rc1, err1 = _urmap(["-map", bad, "-ufi", small_case["ufi"], "-samout", sam, "-batch", "500"])
    rc2, err2 = _urmap(["-map", bad, "-ufi", small_case["ufi"], "-samout", sam2, "-batch", "500"], env={"URMAPX_HOST_TEXT": "1"})
    msg = lambda e: [l for l in e.split("\n") if "FASTQ" in l or "line" in l.lower()]
    assert rc1 == 1 and rc2 == 1 and msg(err1) == msg(err2) and msg(err1), (err1[-500:], err2[-500:])
    trunc = os.path.join(tmp_path, "trunc.fq.gz")
    z = gzip.compress(data, 1)
    open(trunc, "wb").write(z[: len(z) // 2])
    rc, err = _urmap(["-map", trunc, "-ufi", small_case["ufi"], "-samout", sam])
    assert rc == 1, err[-500:]


@pytest.mark.gpu
@pytest.mark.parametrize("allocs", ["0", "1", "5"])
@pytest.mark.parametrize("kind", ["plain", "gz", "pairs"])
def test_cli_text_phase_without_pinned_memory_falls_back_to_the_host(small_case, tmp_path, kind, allocs):
    """No page-locked memory for a chunk (memlock limit, little host memory) ends the device text phase, not the run: the
    host reader and formatter continue from that chunk and the SAM is the same (URMAPX_TEST_PINNED_ALLOCS makes
    the (N+1)-th fresh page-locked allocation of the process fail: N = 0 before the first chunk, later ones mid-file)."""
    from urmap_amd import synth
    f1, f2, osam, sam = (os.path.join(tmp_path, n) for n in ("r1.fq", "r2.fq", "o.sam", "g.sam"))
    if kind == "pairs":
        r1, r2 = synth.make_pairs(815, small_case["genome"], 2500, read_len=150, sub1=0.02, sub2=0.03, ins=0.002, dele=0.002)
        synth.write_fastq(f1, r1)
        synth.write_fastq(f2, r2)
        small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4)
        args = ["-map2", f1, "-reverse", f2]
    else:
        synth.write_fastq(f1, synth.make_reads(814, small_case["genome"], 5000, read_len=150, sub=0.02, ins=0.002, dele=0.002))
        small_case["oracle_index"].map_file_se(f1, osam, threads=4)
        if kind == "gz":
            open(f1 + ".gz", "wb").write(gzip.compress(open(f1, "rb").read(), 1))
            args = ["-map", f1 + ".gz"]
        else:
            args = ["-map", f1]
    want = _file_records(osam)
    for batch in ("100000", "600"):
        rc, err = _urmap(args + ["-ufi", small_case["ufi"], "-samout", sam, "-batch", batch], env={"URMAPX_TEST_PINNED_ALLOCS": allocs})
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want, (batch, err[-600:])


def _urmap_piped(args, feeds, env=None):
    """Runs urmap with FIFOs: feeds = {placeholder: bytes}; every placeholder in args is replaced by a FIFO that a thread
    fills.  A placeholder "-" feeds standard input through a pipe instead."""
    import subprocess
    import tempfile
    import threading
    d = tempfile.mkdtemp()
    argv, threads, stdin_data = [], [], None
    for a in args:
        if a in feeds and a != "-":
            path = os.path.join(d, a.strip("{}") + ".fq")
            os.mkfifo(path)

            def fill(path=path, data=feeds[a]):
                with open(path, "wb") as f:
                    try:
                        f.write(data)
                    except BrokenPipeError:
                        pass
            threads.append(threading.Thread(target=fill))
            argv.append(path)
        else:
            if a == "-" and "-" in feeds:
                stdin_data = feeds["-"]
            argv.append(a)
    p = subprocess.Popen([os.path.join(ROOT, "urmap_amd", "urmap")] + argv, stdin=subprocess.PIPE if stdin_data is not None else subprocess.DEVNULL,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, env={**os.environ, "URMAPX_VERBOSE": "1", **(env or {})})
    for t in threads:
        t.start()
    out, err = p.communicate(stdin_data, timeout=300)
    for t in threads:
        t.join()
    return p.returncode, err.decode()


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["stdin", "fifo"])
def test_cli_piped_input_takes_the_device_text_path(small_case, tmp_path, how):
    """A pipe (FIFO or standard input; myutils.cpp:426-446 opens "-" as stdin) is streamed to the device like a .gz file:
    no host parsing or formatting ("format 0.00"), the oracle's records, at several chunk sizes."""
    from urmap_amd import synth
    reads = synth.make_reads(821, small_case["genome"], 6000, read_len=150, sub=0.02, ins=0.002, dele=0.002, random_frac=0.02)
    fq, osam, sam = (os.path.join(tmp_path, n) for n in ("r.fq", "o.sam", "g.sam"))
    synth.write_fastq(fq, reads)
    small_case["oracle_index"].map_file_se(fq, osam, threads=4)
    data = open(fq, "rb").read()
    want = _file_records(osam)
    name = "-" if how == "stdin" else "{a}"
    for batch in ("100000", "700", "64"):
        rc, err = _urmap_piped(["-map", name, "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch], {name: data})
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want, batch
        assert "format 0.00" in err, err[-600:]


@pytest.mark.gpu
@pytest.mark.parametrize("second", ["fifo", "plain", "gz"])
def test_cli_piped_pairs(small_case, tmp_path, second):
    from urmap_amd import synth
    r1, r2 = synth.make_pairs(822, small_case["genome"], 3000, read_len=150, sub1=0.02, sub2=0.03, ins=0.002, dele=0.002)
    r2 = [(lab + " a longer label", s, q) if k % 7 == 0 else (lab, s, q) for k, (lab, s, q) in enumerate(r2)]
    f1, f2, osam, sam = (os.path.join(tmp_path, n) for n in ("r1.fq", "r2.fq", "o.sam", "g.sam"))
    synth.write_fastq(f1, r1)
    synth.write_fastq(f2, r2)
    small_case["oracle_index"].map_file_pe(f1, f2, osam, threads=4)
    want = _file_records(osam)
    feeds = {"{a}": open(f1, "rb").read()}
    name2 = f2
    if second == "fifo":
        feeds["{b}"] = open(f2, "rb").read()
        name2 = "{b}"
    elif second == "gz":
        name2 = f2 + ".gz"
        open(name2, "wb").write(gzip.compress(open(f2, "rb").read(), 1))
    for batch in ("100000", "512"):
        rc, err = _urmap_piped(["-map2", "{a}", "-reverse", name2, "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch], feeds)
        assert rc == 0, err[-2000:]
        assert _file_records(sam) == want, batch
        assert "format 0.00" in err, err[-600:]


@pytest.mark.gpu
@pytest.mark.parametrize("damage", ["crlf_from_the_middle", "no_final_newline", "bad_record", "blank_lines_at_the_end"])
def test_cli_piped_input_hand_back(small_case, tmp_path, damage):
    """What the device parser hands back cannot be re-read from a pipe: the bytes of that chunk and of every chunk read
    behind it go to the host reader in front of the rest of the pipe.  Same SAM / same message as the host-only run."""
    from urmap_amd import synth
    reads = synth.make_reads(823, small_case["genome"], 4000, read_len=150, sub=0.02, ins=0.002, dele=0.002)
    fq, sam, hsam = (os.path.join(tmp_path, n) for n in ("r.fq", "g.sam", "h.sam"))
    synth.write_fastq(fq, reads)
    data = open(fq, "rb").read()
    lines = data.split(b"\n")
    if damage == "crlf_from_the_middle":
        k = 4 * 2500
        data = b"\n".join(lines[:k]) + b"\n" + b"\r\n".join(lines[k:])
    elif damage == "no_final_newline":
        data = data[:-1]
    elif damage == "bad_record":
        lines[4 * 3100 + 2] = b"-"
        data = b"\n".join(lines)
    else:
        data = data + b"\n\n\n"
    open(fq, "wb").write(data)
    rc0, err0 = _urmap(["-map", fq, "-ufi", small_case["ufi"], "-samout", hsam], env={"URMAPX_HOST_TEXT": "1"})
    for batch in ("100000", "900", "64"):
        rc, err = _urmap_piped(["-map", "{a}", "-ufi", small_case["ufi"], "-samout", sam, "-batch", batch], {"{a}": data})
        assert rc == rc0, (batch, err[-1500:])
        if rc0 == 0:
            assert _file_records(sam) == _file_records(hsam), batch
        else:  # the reference's Die text with the line number; the file name differs (a FIFO), the rest does not
            msg0 = [l for l in err0.splitlines() if "ine " in l and "r.fq" in l]
            msg = [l for l in err.splitlines() if "ine " in l and "a.fq" in l]
            assert msg0 and msg and msg[0].replace("a.fq", "r.fq").split("r.fq")[-1] == msg0[0].split("r.fq")[-1], (err0[-400:], err[-400:])


@pytest.mark.gpu
@pytest.mark.parametrize("name,ufi_gz", [("pe150", "g.ufi.gz"), ("pe120_rep", "r.ufi.gz")])
@pytest.mark.parametrize("form", ["plain", "gz", "fifo"])
def test_cli_map2_tabbedout_with_the_device_text_path(tmp_path, name, ufi_gz, form):
    """-tabbedout no longer sends a -map2 run to the host text stages: the SAM text is made on the device, the tab lines
    by host threads from the chunk's results and pair records (urmapx_text_fetch_pairs).  Reference's golden .tab and
    .sam, at several chunk sizes; the stage report shows that the device wrote the SAM."""
    gold = os.path.join(ROOT, "tests", "golden")
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(gold, ufi_gz), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    d1, d2 = (open(os.path.join(gold, f"{name}_{k}.fq"), "rb").read() for k in (1, 2))
    sam, tab = os.path.join(tmp_path, "out.sam"), os.path.join(tmp_path, "out.tab")
    want_tab = open(os.path.join(gold, name + ".tab"), "rb").read()
    want_sam = [l for l in open(os.path.join(gold, name + ".sam"), "rb").read().split(b"\n") if l]
    for batch in ("100000", "128", "30"):
        if form == "fifo":
            rc, err = _urmap_piped(["-map2", "{a}", "-reverse", "{b}", "-ufi", ufi, "-samout", sam, "-tabbedout", tab, "-batch", batch], {"{a}": d1, "{b}": d2})
        else:
            f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
            if form == "gz":
                f1, f2 = f1 + ".gz", f2 + ".gz"
                open(f1, "wb").write(gzip.compress(d1, 1))
                open(f2, "wb").write(_bgzf(d2))
            else:
                open(f1, "wb").write(d1)
                open(f2, "wb").write(d2)
            rc, err = _urmap(["-map2", f1, "-reverse", f2, "-ufi", ufi, "-samout", sam, "-tabbedout", tab, "-batch", batch])
        assert rc == 0, err[-2000:]
        assert open(tab, "rb").read() == want_tab, batch
        assert [l for l in open(sam, "rb").read().split(b"\n") if l and not l.startswith(b"@PG")] == want_sam, batch
        assert "text on device: 1" in err, err[-600:]


@pytest.mark.gpu
def test_cli_map2_tabbedout_hand_back_mid_file(tmp_path):
    """A CRLF part of the first mate file hands the rest of the run to the host stages: the tab file continues seamlessly."""
    gold = os.path.join(ROOT, "tests", "golden")
    ufi = os.path.join(tmp_path, "x.ufi")
    with gzip.open(os.path.join(gold, "r.ufi.gz"), "rb") as z, open(ufi, "wb") as f:
        f.write(z.read())
    d1, d2 = (open(os.path.join(gold, f"pe120_rep_{k}.fq"), "rb").read() for k in (1, 2))
    l1 = d1.split(b"\n")
    k = 4 * (len(l1) // 8)
    d1 = b"\n".join(l1[:k]) + b"\n" + b"\r\n".join(l1[k:])
    f1, f2 = os.path.join(tmp_path, "r1.fq"), os.path.join(tmp_path, "r2.fq")
    open(f1, "wb").write(d1)
    open(f2, "wb").write(d2)
    sam, tab = os.path.join(tmp_path, "out.sam"), os.path.join(tmp_path, "out.tab")
    want_tab = open(os.path.join(gold, "pe120_rep.tab"), "rb").read()
    for batch in ("100000", "64"):
        rc, err = _urmap(["-map2", f1, "-reverse", f2, "-ufi", ufi, "-samout", sam, "-tabbedout", tab, "-batch", batch])
        assert rc == 0, err[-2000:]
        assert open(tab, "rb").read() == want_tab, batch
