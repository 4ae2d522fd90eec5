"""The product's FASTQ reader (urmapx_fastq_*, the batch form of FASTQSeqSource::GetNextLo, fastqseqsource.cpp:9-116)
on the edge cases the reference's reader defines.  CPU only."""
import gzip
import os

import numpy as np
import pytest

from urmap_amd import api


def _records(n, L=50, seed=0):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        ln = L if L > 0 else int(rng.integers(1, 200))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), ln))
        q = bytes(rng.integers(33, 74, ln, dtype=np.uint8))
        recs.append((f"read{i} extra/1".encode(), s, q))
    return recs


def _text(recs, eol=b"\n", final_eol=True):
    t = b"".join(b"@" + l + eol + s + eol + b"+" + eol + q + eol for l, s, q in recs)
    return t if final_eol else t[:-len(eol)]


def _read_all(path, batch):
    rd = api.FastqReader(path)
    out = []
    while True:
        b = rd.next(batch)
        if b is None:
            break
        labels, bases, offs, quals = b
        for i, lab in enumerate(labels):
            out.append((lab.encode("latin-1"), bases[int(offs[i]):int(offs[i + 1])].tobytes(),
                        quals[int(offs[i]):int(offs[i + 1])].tobytes()))
    rd.close()
    return out


@pytest.mark.parametrize("batch", [1, 7, 1000, 1 << 20])
@pytest.mark.parametrize("variant", ["plain", "crlf", "no_final_eol", "trailing_blank", "ragged", "gz"])
def test_reader_round_trip(tmp_path, batch, variant):
    recs = _records(2500 if batch > 1 else 300, L=0 if variant == "ragged" else 50, seed=3)
    if variant == "crlf":
        data = _text(recs, eol=b"\r\n")
    elif variant == "no_final_eol":
        data = _text(recs, final_eol=False)
    elif variant == "trailing_blank":
        data = _text(recs) + b"\n\n\r\n\n\n"
    else:
        data = _text(recs)
    p = tmp_path / ("r.fq.gz" if variant == "gz" else "r.fq")
    if variant == "gz":
        with gzip.open(p, "wb") as f:
            f.write(data)
    else:
        p.write_bytes(data)
    assert _read_all(str(p), batch) == recs


def test_empty_file_and_missing_file(tmp_path):
    p = tmp_path / "e.fq"
    p.write_bytes(b"")
    assert _read_all(str(p), 10) == []
    with pytest.raises(api.UrmapxError):
        api.FastqReader(str(tmp_path / "nope.fq"))


@pytest.mark.parametrize("bad,msg", [
    (b"@a\nACGT\n+\nIIII\nXa\nACGT\n+\nIIII\n", "expected '@'"),
    (b"@a\nAC1T\n+\nIIII\n", "Invalid sequence letter '1' in FASTQ, line 2 file "),
    (b"@a\nACGT\n+\nIIII\n@b\nAC-T\n+\nIIII\n", "Invalid sequence letter '-' in FASTQ, line 6 file "),
    (b"@a b c\nAC\x01T\n+\nIIII\n", r"Non-printing byte 0x01 in FASTQ sequence line 2 file \S+ label a b c$"),
    (b"@a\nACGT\n+\nIII\n", r"Bad FASTQ record: 4 bases, 3 quals line 4 file \S+ label a$"),
    (b"@a\nACGT\n+\nIIII\n@lab two\nACGTA\n+\nIIII\n", r"Bad FASTQ record: 5 bases, 4 quals line 8 file \S+ label lab two$"),
    (b"@a\nACGT\n+\n", "Unexpected end-of-file in FASTQ file "),
    (b"@a\n", "Unexpected end-of-file in FASTQ file "),
    (b"@a\nACGT\n+\nIIII\n\n@b\nACGT\n+\nIIII\n", "Empty line nr 5 in FASTQ file '"),
    (b"@a\nACGT\n+\nIIII\n\n\r\n\n@b\nACGT\n+\nIIII\n", "Empty line nr 7 in FASTQ file '"),
])
@pytest.mark.parametrize("batch", [1, 100])
def test_reader_errors_use_reference_messages(tmp_path, bad, msg, batch):
    """Message formats of FASTQSeqSource::GetNextLo (fastqseqsource.cpp:31-43, 46-51, 67-84, 95-106); the first, third,
    fourth, sixth and seventh were also compared with the reference binary's stderr (it crashes instead of reporting
    when the problem comes after a good record)."""
    p = tmp_path / "bad.fq"
    p.write_bytes(bad)
    rd = api.FastqReader(str(p))
    with pytest.raises(ValueError, match=msg):
        while rd.next(batch) is not None:
            pass


def test_large_file_crosses_buffer_boundaries(tmp_path):
    # > 64 MiB of input so that the reader's block buffer is refilled several times
    recs = _records(1000, L=150, seed=9)
    data = _text(recs)
    reps = (70 << 20) // len(data) + 1
    p = tmp_path / "big.fq"
    with open(p, "wb") as f:
        for _ in range(reps):
            f.write(data)
    rd = api.FastqReader(str(p))
    n = 0
    while True:
        b = rd.next(300_000)
        if b is None:
            break
        labels, bases, offs, quals = b
        assert np.all(np.diff(offs) == 150)
        k = len(labels)
        for j in (0, k // 2, k - 1):  # record j of this batch is record (n + j) % 1000 of the template
            lab, s, q = recs[(n + j) % 1000]
            assert labels[j].encode() == lab
            assert bases[int(offs[j]):int(offs[j + 1])].tobytes() == s
            assert quals[int(offs[j]):int(offs[j + 1])].tobytes() == q
        n += k
    assert n == reps * 1000
    os.remove(p)


def test_sequence_letter_rule_at_every_offset(tmp_path):
    """The sequence line must be letters only (isalpha, fastqseqsource.cpp:70-77): every A-Z / a-z byte is accepted,
    the bytes just outside the two ranges and bytes >= 0x80 are refused wherever they stand (the check runs eight
    bytes at a time with a tail), and a '\\r' inside the line is dropped, not refused."""
    L = 21
    letters = bytes(range(ord("A"), ord("Z") + 1)) + bytes(range(ord("a"), ord("z") + 1))
    p = tmp_path / "ok.fq"
    p.write_bytes(b"@all\n" + letters + b"\n+\n" + b"I" * len(letters) + b"\n@cr\nAC\rGT\r\n+\nIIII\r\n")
    got = _read_all(str(p), 10)
    assert got == [(b"all", letters, b"I" * len(letters)), (b"cr", b"ACGT", b"IIII")]
    for badbyte in (b"@", b"[", b"`", b"{", b"0", b" ", b"\xc1", b"\xe1", b"\x8d", b"\x00"):
        for at in range(L):
            seq = b"A" * at + badbyte + b"C" * (L - 1 - at)
            p = tmp_path / "bad.fq"
            p.write_bytes(b"@r\n" + seq + b"\n+\n" + b"I" * L + b"\n")
            rd = api.FastqReader(str(p))
            with pytest.raises(ValueError, match="Invalid sequence letter '|Non-printing byte 0x"):
                rd.next(10)
            rd.close()


def test_reader_takes_a_fifo(tmp_path):
    """A named pipe (e.g. `-map <(zcat reads.fq.gz)`) cannot be read at offsets: it is consumed front to back."""
    import threading
    recs = _records(5000, L=0, seed=11)
    data = _text(recs)
    p = str(tmp_path / "pipe.fq")
    os.mkfifo(p)

    def feed():
        with open(p, "wb") as f:
            for i in range(0, len(data), 70001):
                f.write(data[i:i + 70001])
    t = threading.Thread(target=feed)
    t.start()
    try:
        assert _read_all(p, 777) == recs
    finally:
        t.join()


def test_open_failures_and_corrupt_gzip(tmp_path):
    """OpenStdioFile / OpenGzipFile / ReadGzipFile messages (myutils.cpp:426-446, gzipfileio.cpp:8-22)."""
    for name in ("nope.fq", "nope.fq.gz"):  # the C ABI reports the class (URMAPX_E_IO); the command line prints the message
        with pytest.raises(api.UrmapxError) as e:
            api.FastqReader(str(tmp_path / name))
        assert e.value.code == api.E_IO
    recs = _records(3000, L=100, seed=5)
    p = tmp_path / "bad.fq.gz"
    blob = bytearray(gzip.compress(_text(recs)))
    for i in range(len(blob) // 2, len(blob) // 2 + 64):
        blob[i] ^= 0x5A  # damage the deflate stream
    p.write_bytes(bytes(blob))
    rd = api.FastqReader(str(p))
    with pytest.raises(ValueError, match="Error reading gzip file|Bad|Invalid|Non-printing|Unexpected"):
        while rd.next(500) is not None:
            pass


@pytest.mark.parametrize("batch", [50_000, 333_333])
def test_record_length_shifts_midway(tmp_path, batch):
    """The reader sizes its reads from the line length seen so far; a file whose records get six times longer (and
    shorter again) in the middle must still come back record for record, across buffer refills and compactions."""
    short = _records(997, L=40, seed=21)
    long_ = _records(991, L=300, seed=22)
    blocks = [(short, 300), (long_, 200), (short, 150), (long_, 60)]  # ~110 MB in all
    p = tmp_path / "shift.fq"
    order = []
    with open(p, "wb") as f:
        for recs, reps in blocks:
            data = _text(recs)
            for _ in range(reps):
                f.write(data)
            order.append((recs, reps * len(recs)))
    rd = api.FastqReader(str(p))
    seq = []  # (template, index) of every record in file order, generated lazily
    def expected(i):
        for recs, cnt in order:
            if i < cnt:
                return recs[i % len(recs)]
            i -= cnt
        raise IndexError
    n = 0
    total = sum(c for _, c in order)
    while True:
        b = rd.next(batch)
        if b is None:
            break
        labels, bases, offs, quals = b
        k = len(labels)
        lens = np.diff(offs).astype(np.int64)
        for j in list(range(0, k, 4099)) + [k - 1]:
            lab, s, q = expected(n + j)
            assert labels[j].encode() == lab and lens[j] == len(s)
            assert bases[int(offs[j]):int(offs[j + 1])].tobytes() == s
            assert quals[int(offs[j]):int(offs[j + 1])].tobytes() == q
        n += k
    assert n == total
    os.remove(p)
