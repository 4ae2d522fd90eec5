"""UFIndex::Validate (ufindex.cpp:611-658) as a device pass over the resident table (urmapx_index_validate, `urmap -ufi_validate`):
green on tables written by the reference binary and by the oracle, red on every kind of damage the pass names."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "urmap_amd", "urmap")


def _gunzip(name, tmp_path):
    p = os.path.join(tmp_path, name[:-3])
    with gzip.open(os.path.join(GOLD, name), "rb") as z, open(p, "wb") as f:
        f.write(z.read())
    return p


def _parts(path):
    """(bytes of the file, offset of the slot table, slot count, sequence size) of a .ufi file (ufindexio.cpp:14-49)"""
    raw = bytearray(open(path, "rb").read())
    magic, w, maxix, sds, slots, nseq = struct.unpack_from("<IIIIQI", raw, 0)
    assert magic == 0x55464931
    off = 28
    for _ in range(nseq):
        _, _, n = struct.unpack_from("<III", raw, off)
        off += 12 + n
    off += 4  # UFI2
    return raw, off, slots, sds


def _tallies(raw, off, slots):
    return np.frombuffer(bytes(raw[off:off + 5 * slots]), np.uint8)[0::5]


@pytest.mark.parametrize("name", ["g.ufi.gz", "r.ufi.gz"])
def test_reference_written_tables_validate(tmp_path, name):
    """the reference binary's own -make_ufi output (tests/golden) passes, and the counts are the table's"""
    from urmap_amd import api
    ufi = _gunzip(name, str(tmp_path))
    raw, off, slots, sds = _parts(ufi)
    t = _tallies(raw, off, slots)
    ok, rep = api.Index.open(ufi).upload(0).validate()
    assert ok, rep
    assert rep["slots"] == slots
    assert rep["used"] == int((t != 0).sum()) == rep["reached"]
    assert rep["heads"] == int((t >= 128).sum())
    # a position per chain link; the middle slot of a long link holds the position of the link before it
    n_long = int(((t == 253) | (t == 125)).sum())
    assert rep["positions"] + n_long // 2 >= rep["used"] - n_long and rep["positions"] <= rep["used"]
    assert rep["first_bad_slot"] == 0xFFFFFFFFFFFFFFFF and rep["bad_hash"] == rep["bad_pos"] == rep["bad_link"] == rep["bad_len"] == 0


def test_oracle_built_table_validates(small_case):
    from urmap_amd import api
    ok, rep = api.Index.open(small_case["ufi"]).upload(0).validate()
    assert ok and rep["used"] == rep["reached"] and rep["positions"] > 100000, rep


def _damaged(tmp_path, kind):
    ufi = _gunzip("g.ufi.gz", str(tmp_path))
    raw, off, slots, sds = _parts(ufi)
    t = _tallies(raw, off, slots)
    if kind == "hash":  # a unique k-mer's position moved by one base
        s = int(np.nonzero(t == 255)[0][1000])
        pos = struct.unpack_from("<I", raw, off + 5 * s + 1)[0]
        struct.pack_into("<I", raw, off + 5 * s + 1, pos + 1)
    elif kind == "pos":  # a position beyond the sequence store
        s = int(np.nonzero(t == 254)[0][10])
        struct.pack_into("<I", raw, off + 5 * s + 1, sds + 5)
    elif kind == "link":  # a chain link freed under its chain
        s = int(np.nonzero((t > 0) & (t < 125))[0][50])
        raw[off + 5 * s] = 0
    elif kind == "mine":  # a link that claims to head a row of its own
        s = int(np.nonzero((t > 0) & (t < 125))[0][77])
        raw[off + 5 * s] |= 128
    elif kind == "orphan":  # a used slot no chain leads to
        s = int(np.nonzero(t == 0)[0][123])
        raw[off + 5 * s] = 127
    p = os.path.join(str(tmp_path), kind + ".ufi")
    open(p, "wb").write(raw)
    return p, s


@pytest.mark.parametrize("kind,field", [("hash", "bad_hash"), ("pos", "bad_pos"), ("link", "bad_link"), ("mine", "bad_link"), ("orphan", None)])
def test_damage_is_found(tmp_path, kind, field):
    from urmap_amd import api
    p, s = _damaged(tmp_path, kind)
    ok, rep = api.Index.open(p).upload(0).validate()
    assert not ok, rep
    if field:
        assert rep[field] >= 1, rep
        assert rep["first_bad_slot"] != 0xFFFFFFFFFFFFFFFF
    if kind in ("hash", "pos"):
        assert rep["first_bad_slot"] == s and rep[field] == 1
    if kind == "orphan":
        assert rep["used"] == rep["reached"] + 1 and rep["bad_hash"] == 0


def test_command_line(tmp_path):
    """urmap -ufi_validate: exit 0 and the counts on a good table; the reference's message and exit 1 on a bad one"""
    good = _gunzip("g.ufi.gz", str(tmp_path))
    r = subprocess.run([EXE, "-ufi_validate", good], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr
    assert b"positions re-hashed" in r.stderr
    bad, _ = _damaged(tmp_path, "hash")
    r = subprocess.run([EXE, "-ufi_validate", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1
    assert b"WordToSlot != Slot" in r.stderr
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "urmap")):  # the reference's own verdicts on the same two files
        ref = os.path.join(ROOT, "oracle", "_ref", "urmap")
        assert subprocess.run([ref, "-ufi_validate", good], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 0
        rb = subprocess.run([ref, "-ufi_validate", bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert rb.returncode != 0 and b"WordToSlot != Slot" in rb.stderr + rb.stdout
