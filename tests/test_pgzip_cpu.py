"""The .gz reader of urmapx_map_files (urmap_amd/csrc/pgzip.cpp: a gzip stream inflated by several threads, segments decoded with
the window in front of them unknown) against zlib: the same bytes for every kind of stream zlib writes, the same verdict on
damaged files.  Host code only -- no GPU involved (urmapx_gunzip_file)."""
import gzip
import io
import os
import zlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def fastq():
    rng = np.random.default_rng(5)
    n, L = 60000, 150
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
    quals = (rng.integers(0, 40, (n, L)) + 33).astype(np.uint8)
    out = bytearray()
    for i in range(n):
        out += b"@read%09d/1 extra\n" % i + bases[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n"
    return bytes(out)  # 19.6 MB


def _z(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None):
    co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    out = b""
    if flush_every:
        for i in range(0, len(data), flush_every):
            out += co.compress(data[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH)
    else:
        out = co.compress(data)
    return out + co.flush()


def _run(tmp_path, name, gz_bytes, threads=8, segment="65536", monkeypatch=None):
    from urmap_amd import api
    p = os.path.join(tmp_path, name + ".gz")
    open(p, "wb").write(gz_bytes)
    if monkeypatch is not None:
        if segment:
            monkeypatch.setenv("URMAPX_PGZIP_SEGMENT", segment)  # small segments: many junctions in a small file
        else:
            monkeypatch.delenv("URMAPX_PGZIP_SEGMENT", raising=False)
    st = api.gunzip_file(p, p + ".out", threads)
    return open(p + ".out", "rb").read(), st


CASES = {
    "level1": lambda t: gzip.compress(t, 1),
    "level6": lambda t: gzip.compress(t, 6),
    "level9": lambda t: gzip.compress(t, 9),
    "members": lambda t: b"".join(gzip.compress(t[i:i + 3_000_000], (1, 6, 9)[(i // 3_000_000) % 3]) for i in range(0, len(t), 3_000_000)),
    "trailing_zeros": lambda t: gzip.compress(t, 6) + b"\0" * 777,
    "trailing_garbage": lambda t: gzip.compress(t, 6) + b"not a gzip member",
    "full_flush": lambda t: _z(t, 6, flush_every=250_000),
    "fixed_huffman": lambda t: _z(t[:4_000_000], 6, zlib.Z_FIXED),
    "stored": lambda t: _z(t[:4_000_000], 0),
    "huffman_only": lambda t: _z(t[:4_000_000], 6, zlib.Z_HUFFMAN_ONLY),
    "rle": lambda t: _z(t[:4_000_000], 6, zlib.Z_RLE),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_streams_zlib_writes(tmp_path, fastq, case, monkeypatch):
    gz = CASES[case](fastq)
    want = gzip.decompress(gz) if case not in ("trailing_garbage",) else fastq
    got, st = _run(str(tmp_path), case, gz, monkeypatch=monkeypatch)
    assert got == want
    assert st[0] == len(want) and st[1] == len(want) and st[2] == 0  # all of it by the parallel road


def test_default_segments_and_thread_counts(tmp_path, fastq, monkeypatch):
    gz = gzip.compress(fastq, 6)
    got, st = _run(str(tmp_path), "seg2m", gz, segment=None, monkeypatch=monkeypatch)
    assert got == fastq and st[1] == len(fastq)
    got, st = _run(str(tmp_path), "one", gz, threads=1, monkeypatch=monkeypatch)
    assert got == fastq and st[2] == len(fastq)  # one thread: zlib
    got, st = _run(str(tmp_path), "three", gz, threads=3, monkeypatch=monkeypatch)
    assert got == fastq and st[1] == len(fastq)
    got, st = _run(str(tmp_path), "seg4k", gzip.compress(fastq[:2_000_000], 6), segment="4096", monkeypatch=monkeypatch)
    assert got == fastq[:2_000_000]


def test_header_fields_tiny_and_empty(tmp_path, fastq, monkeypatch):
    bio = io.BytesIO()
    with gzip.GzipFile(filename="some_reads.fastq", mode="wb", fileobj=bio, compresslevel=6, mtime=12345) as f:
        f.write(fastq[:3_000_000])
    got, _ = _run(str(tmp_path), "fname", bio.getvalue(), monkeypatch=monkeypatch)
    assert got == fastq[:3_000_000]
    tiny = b"@r\nACGT\n+\nIIII\n"
    got, st = _run(str(tmp_path), "tiny", gzip.compress(tiny), monkeypatch=monkeypatch)
    assert got == tiny and st[2] == len(tiny)
    got, st = _run(str(tmp_path), "empty", gzip.compress(b""), monkeypatch=monkeypatch)
    assert got == b"" and st[0] == 0


def test_text_that_is_not_fastq_and_bytes_that_are_not_text(tmp_path, fastq, monkeypatch):
    """no block start is found where the text is not text: the segment in front decodes on through it -- same bytes"""
    rnd = np.random.default_rng(1).integers(0, 256, 1_500_000, dtype=np.uint8).tobytes()
    for name, data in (("binary", rnd), ("mixed", fastq[:2_000_000] + rnd[:400_000] + fastq[2_000_000:4_000_000]),
                       ("runs", b"".join(b"@r%08d\n" % i + b"ACGT" * 37 + b"AC\n+\n" + b"I" * 150 + b"\n" for i in range(20000)))):
        got, _ = _run(str(tmp_path), name, gzip.compress(data, 6), monkeypatch=monkeypatch)
        assert got == data, name


def test_damaged_files_fail_as_with_zlib(tmp_path, fastq, monkeypatch):
    from urmap_amd import api
    gz = gzip.compress(fastq, 6)
    bad = bytearray(gz)
    bad[len(bad) // 2] ^= 0x55
    for name, data in (("truncated", gz[:len(gz) // 2]), ("corrupt", bytes(bad)), ("bad_crc", gz[:-8] + b"\0\0\0\0" + gz[-4:])):
        with pytest.raises(api.UrmapxError) as e:
            _run(str(tmp_path), name, data, monkeypatch=monkeypatch)
        assert e.value.code == api.E_FORMAT
        with pytest.raises(Exception):
            gzip.decompress(data)
    with pytest.raises(api.UrmapxError):
        _run(str(tmp_path), "notgz", b"@r\nACGT\n+\nIIII\n" * 10, monkeypatch=monkeypatch)


def test_vector_paths_are_in_use_and_agree_with_the_plain_loops(tmp_path, fastq):
    """Symbols -> bytes 32 at a time (AVX2) and CRC-32 by carry-less multiplication (PCLMULQDQ) are what runs on a host that has the
    instructions -- the CRC routine only after reproducing zlib's crc32 on its self-test, so a wrong constant would show here as a
    missing flag, not as a wrong checksum -- and URMAPX_PGZIP_NO_SIMD=1 (a process of its own: the choice is made once) gives the same bytes."""
    import subprocess
    import sys
    from urmap_amd import api
    flags = open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    got = api.lib().urmapx_pgzip_simd()
    assert bool(got & 1) == ("avx2" in flags)
    assert bool(got & 2) == ("pclmulqdq" in flags and "sse4_1" in flags)
    gz = os.path.join(tmp_path, "v.gz")
    open(gz, "wb").write(gzip.compress(fastq, 6))
    code = ("import sys; sys.path.insert(0, %r); from urmap_amd import api; print(api.lib().urmapx_pgzip_simd(), api.gunzip_file(%r, %r, 8))"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), gz, gz + ".plain"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, URMAPX_PGZIP_NO_SIMD="1", URMAPX_PGZIP_SEGMENT="65536"), capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("0 "), (r.stdout, r.stderr[-500:])
    assert open(gz + ".plain", "rb").read() == fastq
