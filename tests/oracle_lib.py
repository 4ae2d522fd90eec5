"""ctypes binding of oracle/liburmap_oracle.so (the CPU checker; test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liburmap_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "urmap")


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "mismatch_score", "gap_open_score", "gap_ext_score", "min_hsp_score_pct",
        "term_hsp_score_pct_phase3", "xdrop", "max_penalty", "xphase1", "xphase3", "xphase4")] + [
        ("band_radius", C.c_uint)]


class Result(C.Structure):
    _fields_ = [("dbpos", C.c_uint32), ("seq_index", C.c_uint32), ("coord", C.c_uint32),
                ("score", C.c_int32), ("second", C.c_int32), ("mapq", C.c_uint32),
                ("hit_count", C.c_uint32), ("hsp_count", C.c_uint32), ("plus", C.c_uint8),
                ("exit_phase", C.c_uint8), ("path_len", C.c_uint16), ("path_off", C.c_uint32)]


RESULT_DTYPE = np.dtype([("dbpos", "<u4"), ("seq_index", "<u4"), ("coord", "<u4"), ("score", "<i4"),
                         ("second", "<i4"), ("mapq", "<u4"), ("hit_count", "<u4"), ("hsp_count", "<u4"),
                         ("plus", "u1"), ("exit_phase", "u1"), ("path_len", "<u2"), ("path_off", "<u4")])
assert RESULT_DTYPE.itemsize == C.sizeof(Result)

PAIR_INFO_DTYPE = np.dtype([("top_db", "<u4", 2), ("second_db", "<u4", 2), ("top_score", "<i2", 2),
                            ("second_score", "<i2", 2), ("top_plus", "u1", 2), ("second_plus", "u1", 2)])
assert PAIR_INFO_DTYPE.itemsize == 28

COUNTER_NAMES = ("n_reads", "n_getblob", "n_rowcalls", "n_rowhop", "n_extend", "n_extbases",
                 "n_alignhsp", "n_viterbi", "n_dpcells", "n_dptarget", "n_qbases", "n_scan", "n_extscan", "n_scan_vit", "n_scan_hits")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in COUNTER_NAMES]

    def asdict(self):
        return {n: int(getattr(self, n)) for n in COUNTER_NAMES}


def build_oracle():
    """make -C oracle oracle (and ref when /root/reference exists)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-j8", "-C", ORACLE_DIR, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "urmap_oracle.cpp")):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    L = C.CDLL(LIB_PATH)
    vp, cp, u32, u64 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64
    L.uo_params_for_method.argtypes = [C.c_uint, C.POINTER(Params)]
    L.uo_index_load.restype = vp
    L.uo_index_load.argtypes = [cp, C.POINTER(cp)]
    L.uo_index_wrap.restype = vp
    L.uo_index_wrap.argtypes = [u32, u32, u64, vp, vp, u32, u32, vp, vp, cp]
    L.uo_index_build.restype = vp
    L.uo_index_build.argtypes = [cp, u32, u32, u64, C.POINTER(cp)]
    L.uo_index_free.argtypes = [vp]
    L.uo_index_save.argtypes = [vp, cp]
    for name, rt in (("word_length", u32), ("max_ix", u32), ("slot_count", u64), ("seqdata_size", u32),
                     ("blob", vp), ("seqdata", vp), ("seq_count", u32)):
        f = getattr(L, "uo_index_" + name)
        f.restype = rt
        f.argtypes = [vp]
    L.uo_index_label.restype = cp
    L.uo_index_label.argtypes = [vp, u32]
    L.uo_index_seq_length.restype = u32
    L.uo_index_seq_length.argtypes = [vp, u32]
    L.uo_index_seq_offset.restype = u32
    L.uo_index_seq_offset.argtypes = [vp, u32]
    L.uo_slots_vec.argtypes = [vp, vp, u32, vp]
    L.uo_revcomp.argtypes = [vp, u32, vp]
    L.uo_get_row.restype = C.c_uint
    L.uo_get_row.argtypes = [vp, u64, vp]
    L.uo_viterbi.restype = C.c_float
    L.uo_viterbi.argtypes = [C.POINTER(Params), vp, C.c_uint, vp, C.c_uint, C.c_int, C.c_int, vp]
    L.uo_map_se.argtypes = [vp, C.POINTER(Params), vp, vp, u32, C.c_int, vp, C.POINTER(vp), C.POINTER(Counters)]
    L.uo_map_pe.argtypes = [vp, C.POINTER(Params), vp, vp, u32, C.c_int, vp, C.POINTER(vp), C.POINTER(Counters)]
    L.uo_map_pe_opts.argtypes = [vp, C.POINTER(Params), vp, vp, u32, C.c_int, C.c_int, vp, C.POINTER(vp), C.POINTER(Counters)]
    L.uo_map_pe_info.argtypes = [vp, C.POINTER(Params), vp, vp, u32, C.c_int, C.c_int, vp, C.POINTER(vp), vp, vp]
    L.uo_free.argtypes = [vp]
    L.uo_sam_se.restype = C.c_size_t
    L.uo_sam_se.argtypes = [vp, vp, cp, cp, vp, vp, u32, vp]
    L.uo_map_file_se.argtypes = [vp, C.POINTER(Params), cp, cp, C.c_int, C.POINTER(Counters)]
    L.uo_map_file_pe.argtypes = [vp, C.POINTER(Params), cp, cp, cp, C.c_int, C.c_int, C.POINTER(Counters)]
    L.uo_map_file_pe_tab.argtypes = [vp, C.POINTER(Params), cp, cp, cp, cp, C.c_int, C.c_int, C.POINTER(Counters)]
    _lib = L
    return L


def params(method=6):
    p = Params()
    assert lib().uo_params_for_method(method, C.byref(p)) == 0
    return p


class Index:
    def __init__(self, handle, keep=()):
        if not handle:
            raise RuntimeError("oracle index handle is NULL")
        self.h = C.c_void_p(handle)
        self._keep = keep

    @classmethod
    def load(cls, path):
        err = C.c_char_p()
        h = lib().uo_index_load(path.encode(), C.byref(err))
        if not h:
            raise RuntimeError(f"uo_index_load({path}): {err.value}")
        return cls(h)

    @classmethod
    def build(cls, fasta, slots, word_length=24, max_ix=32):
        err = C.c_char_p()
        h = lib().uo_index_build(fasta.encode(), word_length, max_ix, slots, C.byref(err))
        if not h:
            raise RuntimeError(f"uo_index_build: {err.value}")
        return cls(h)

    @classmethod
    def wrap(cls, word_length, max_ix, slot_count, blob, seqdata, seq_lengths, offsets, labels):
        """blob/seqdata: contiguous uint8 numpy arrays (kept alive by this object)."""
        sl = np.ascontiguousarray(seq_lengths, dtype=np.uint32)
        of = np.ascontiguousarray(offsets, dtype=np.uint32)
        lab = b"".join(l.encode() + b"\0" for l in labels)
        h = lib().uo_index_wrap(word_length, max_ix, slot_count, blob.ctypes.data, seqdata.ctypes.data,
                                len(seqdata), len(labels), sl.ctypes.data, of.ctypes.data, lab)
        return cls(h, keep=(blob, seqdata, sl, of, lab))

    def save(self, path):
        assert lib().uo_index_save(self.h, path.encode()) == 0

    @property
    def word_length(self): return lib().uo_index_word_length(self.h)
    @property
    def max_ix(self): return lib().uo_index_max_ix(self.h)
    @property
    def slot_count(self): return lib().uo_index_slot_count(self.h)
    @property
    def seqdata_size(self): return lib().uo_index_seqdata_size(self.h)

    def blob(self):
        n = 5 * self.slot_count
        return np.ctypeslib.as_array(C.cast(lib().uo_index_blob(self.h), C.POINTER(C.c_uint8)), shape=(n,))

    def seqdata(self):
        return np.ctypeslib.as_array(C.cast(lib().uo_index_seqdata(self.h), C.POINTER(C.c_uint8)),
                                     shape=(self.seqdata_size,))

    def directory(self):
        n = lib().uo_index_seq_count(self.h)
        return [(lib().uo_index_label(self.h, i).decode(), lib().uo_index_seq_length(self.h, i),
                 lib().uo_index_seq_offset(self.h, i)) for i in range(n)]

    def slots_vec(self, seq: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.full(len(seq), np.iinfo(np.uint64).max, dtype=np.uint64)
        lib().uo_slots_vec(self.h, seq.ctypes.data, len(seq), out.ctypes.data)
        return out[: max(0, len(seq) - self.word_length + 1)]

    def get_row(self, slot):
        pv = np.zeros(self.max_ix + 1, dtype=np.uint32)
        k = lib().uo_get_row(self.h, int(slot), pv.ctypes.data)
        return pv[:k].copy()

    def map_se(self, bases: np.ndarray, offs: np.ndarray, method=6, threads=1, p=None):
        """-> (results structured array, list of path strings, counters dict); p: parameters other than the method's"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        arena = C.c_void_p()
        cnt = Counters()
        p = p if p is not None else params(method)
        rc = lib().uo_map_se(self.h, C.byref(p), bases.ctypes.data, offs.ctypes.data, n, threads,
                             res.ctypes.data, C.byref(arena), C.byref(cnt))
        assert rc == 0
        paths = [C.string_at(arena.value + int(r["path_off"])).decode() for r in res]
        lib().uo_free(arena)
        return res, paths, cnt.asdict()

    def map_pe(self, bases: np.ndarray, offs: np.ndarray, threads=1, veryfast=False, p=None):
        """pairs interleaved (reads 2i, 2i+1) -> (results[2*npairs], paths, counters)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        arena = C.c_void_p()
        cnt = Counters()
        p = p if p is not None else params(6)
        rc = lib().uo_map_pe_opts(self.h, C.byref(p), bases.ctypes.data, offs.ctypes.data, n // 2, threads, int(veryfast),
                                  res.ctypes.data, C.byref(arena), C.byref(cnt))
        assert rc == 0
        paths = [C.string_at(arena.value + int(r["path_off"])).decode() for r in res]
        lib().uo_free(arena)
        return res, paths, cnt.asdict()

    def map_pe_info(self, bases: np.ndarray, offs: np.ndarray, threads=1, veryfast=False):
        """as map_pe, plus the per-pair record State2::OutputTab2 reads (layout of the product's urmapx_pair_info)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        info = np.zeros(n // 2, dtype=PAIR_INFO_DTYPE)
        arena = C.c_void_p()
        p = params(6)
        rc = lib().uo_map_pe_info(self.h, C.byref(p), bases.ctypes.data, offs.ctypes.data, n // 2, threads, int(veryfast),
                                  res.ctypes.data, C.byref(arena), None, info.ctypes.data)
        assert rc == 0
        paths = [C.string_at(arena.value + int(r["path_off"])).decode() for r in res]
        lib().uo_free(arena)
        return res, paths, info

    def map_file_se(self, fastq, sam, method=6, threads=1):
        cnt = Counters()
        p = params(method)
        rc = lib().uo_map_file_se(self.h, C.byref(p), fastq.encode(), sam.encode(), threads, C.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"uo_map_file_se rc={rc}")
        return cnt.asdict()

    def map_file_pe_tab(self, fq1, fq2, sam, tab, threads=1, veryfast=False):
        """sam / tab may be None; tab gets State2::OutputTab2's lines."""
        cnt = Counters()
        p = params(6)
        rc = lib().uo_map_file_pe_tab(self.h, C.byref(p), fq1.encode(), fq2.encode(), sam.encode() if sam else None,
                                      tab.encode() if tab else None, threads, int(veryfast), C.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"uo_map_file_pe_tab rc={rc}")
        return cnt.asdict()

    def map_file_pe(self, fq1, fq2, sam, threads=1, veryfast=False):
        cnt = Counters()
        p = params(6)
        rc = lib().uo_map_file_pe(self.h, C.byref(p), fq1.encode(), fq2.encode(), sam.encode(), threads,
                                  int(veryfast), C.byref(cnt))
        if rc != 0:
            raise RuntimeError(f"uo_map_file_pe rc={rc}")
        return cnt.asdict()

    def __del__(self):
        try:
            lib().uo_index_free(self.h)
        except Exception:
            pass


def viterbi(A: bytes, B: bytes, left: bool, right: bool, method=6):
    p = params(method)
    a = np.frombuffer(A, dtype=np.uint8)
    b = np.frombuffer(B, dtype=np.uint8)
    out = C.create_string_buffer(len(A) + len(B) + 2)
    s = lib().uo_viterbi(C.byref(p), a.ctypes.data if len(a) else None, len(a), b.ctypes.data if len(b) else None,
                         len(b), int(left), int(right), out)
    return float(s), out.value.decode()


def have_ref():
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def run_ref(args, cwd=None):
    """Run the reference binary (oracle/_ref/urmap); raises on failure."""
    r = subprocess.run([REF_BIN] + list(args), cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        raise RuntimeError(f"urmap {' '.join(args)} failed: {r.stderr.decode()[-2000:]}")
    return r


def ufi_header(path):
    """(word_length, max_ix, seqdata_size, slot_count) from a .ufi file."""
    import struct
    with open(path, "rb") as f:
        magic, w, maxix, sds, slots = struct.unpack("<IIIIQ", f.read(24))
    assert magic == 0x55464931
    return w, maxix, sds, slots


def sam_records(path):
    with open(path, "rb") as f:
        return [l for l in f.read().split(b"\n") if l and not l.startswith(b"@PG")]
