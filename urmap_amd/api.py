"""ctypes binding of liburmapx.so (include/urmapx.h) -- the host-side mirror of the reference seam
State1::SetMethod / SetUFI / Search (map.cpp:11-25) and UFIndex::FromFile (ufindexio.cpp:51-115).

Nothing here computes on the CPU: every call goes to the HIP library, and a missing or unloadable
library is a hard error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("URMAPX_LIB") or os.path.join(_HERE, "liburmapx.so")  # URMAPX_LIB: A/B builds

MAX_QL = 1024
MAX_PATH_OPS = int(os.environ.get("URMAPX_MAX_PATH_OPS_OVERRIDE", 96))
E_IO, E_FORMAT, E_NOMEM, E_NODEVICE, E_ARG = -1, -2, -3, -4, -5
E_UNSUPPORTED = -6

RESULT_DTYPE = np.dtype([("dbpos", "<u4"), ("seq_index", "<u4"), ("coord", "<u4"), ("score", "<i2"),
                         ("second", "<i2"), ("mapq", "u1"), ("plus", "u1"), ("exit_phase", "u1"),
                         ("status", "u1"), ("hit_count", "<u2"), ("path_nops", "<u2"), ("path_off", "<u4")])
assert RESULT_DTYPE.itemsize == 28

EXPORTS = (
    "urmapx_params_for_method", "urmapx_index_open", "urmapx_index_wrap_host", "urmapx_index_wrap_device",
    "urmapx_index_upload", "urmapx_index_replicate", "urmapx_index_close", "urmapx_index_chain_row_bytes", "urmapx_index_validate", "urmapx_index_word_length", "urmapx_index_max_ix",
    "urmapx_index_slot_count", "urmapx_index_seqdata_size", "urmapx_index_seq_count", "urmapx_index_label",
    "urmapx_index_seq_length", "urmapx_index_seq_offset", "urmapx_ctx_create", "urmapx_ctx_destroy",
    "urmapx_map_se", "urmapx_map_se_device", "urmapx_ctx_sync", "urmapx_ctx_last_kernel_ms",
    "urmapx_seed_probe", "urmapx_seed_probe_device", "urmapx_viterbi_batch", "urmapx_strerror", "urmapx_device_arch",
    "urmapx_make_ufi", "urmapx_make_ufi_opts", "urmapx_build_slots", "urmapx_make_ufi_gpu", "urmapx_build_slots_gpu", "urmapx_sam_se", "urmapx_sam_header_sq", "urmapx_ctx_phase_cycles", "urmapx_ctx_read_cycles", "urmapx_ctx_stage_ms", "urmapx_ctx_phase3", "urmapx_ctx_round_ms", "urmapx_ctx_dp_rounds", "urmapx_ctx_dp_stats", "urmapx_map_pe", "urmapx_sam_pe", "urmapx_map_pe_device", "urmapx_ctx_set_pe_veryfast",
    "urmapx_gunzip_file", "urmapx_fastq_open", "urmapx_fastq_next", "urmapx_fastq_error", "urmapx_fastq_close",
    "urmapx_ctx_gather_microbench", "urmapx_map_files", "urmapx_host_pool_trim", "urmapx_text_create", "urmapx_text_destroy", "urmapx_text_map_se", "urmapx_text_map_pe", "urmapx_pgzip_simd", "urmapx_index_open_device", "urmapx_text_fetch_sam", "urmapx_text_set_deferred", "urmapx_text_wait", "urmapx_text_fetch_pairs", "urmapx_ctx_set_pair_info", "urmapx_ctx_get_pair_info", "urmapx_tab_pe",
    "urmapx_checksum_device", "urmapx_index_checksum", "urmapx_index_layout_checksum",
)


PAIR_INFO_DTYPE = np.dtype([("top_db", "<u4", 2), ("second_db", "<u4", 2), ("top_score", "<i2", 2), ("second_score", "<i2", 2),
                            ("top_plus", "u1", 2), ("second_plus", "u1", 2)])
assert PAIR_INFO_DTYPE.itemsize == 28


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mismatch_score", "gap_open_score", "gap_ext_score", "min_hsp_score_pct",
        "term_hsp_score_pct_phase3", "xdrop", "max_penalty", "xphase1", "xphase3", "xphase4")] + [
        ("band_radius", C.c_uint32)]


class MapOptions(C.Structure):
    _fields_ = [("first_gpu", C.c_int), ("gpus", C.c_int), ("streams", C.c_int), ("host_threads", C.c_int), ("batch", C.c_uint32),
                ("veryfast", C.c_int), ("minq", C.c_uint), ("cmdline", C.c_char_p), ("sam_shards", C.c_int), ("discard_sam", C.c_int)]


class MapReport(C.Structure):
    _fields_ = [("reads", C.c_uint64), ("mapped_q", C.c_uint64), ("mapped_lowq", C.c_uint64), ("unmapped", C.c_uint64),
                ("unsupported", C.c_uint64), ("seconds", C.c_double), ("parse_s", C.c_double), ("gpu_s", C.c_double),
                ("format_s", C.c_double), ("write_s", C.c_double), ("host_threads", C.c_int), ("lanes", C.c_int),
                ("write_threads", C.c_int), ("text_on_device", C.c_int), ("input_bytes", C.c_uint64), ("medium", C.c_char * 24),
                ("dev_h2d_s", C.c_double), ("dev_parse_s", C.c_double), ("dev_map_s", C.c_double), ("dev_format_s", C.c_double),
                ("dev_d2h_s", C.c_double), ("shards", C.c_int), ("placement", C.c_char * 256), ("shard_scan_s", C.c_double),
                ("dev_map_search_s", C.c_double), ("dev_map_dp_s", C.c_double), ("map_enqueue_s", C.c_double),
                ("alloc_dev_s", C.c_double), ("alloc_pinned_s", C.c_double), ("alloc_dev_calls", C.c_uint32), ("alloc_pinned_calls", C.c_uint32)]


class ValidateReport(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("slots", "heads", "positions", "used", "reached", "bad_hash", "bad_pos", "bad_link", "bad_len",
                                          "first_bad_slot")] + [("seconds", C.c_double)]


class TextReport(C.Structure):
    _fields_ = [("records", C.c_uint32), ("reason", C.c_uint32), ("sam_bytes", C.c_uint64), ("mapped_q", C.c_uint64),
                ("mapped_lowq", C.c_uint64), ("unmapped", C.c_uint64), ("unsupported", C.c_uint64),
                ("ms_h2d", C.c_float), ("ms_parse", C.c_float), ("ms_map", C.c_float), ("ms_format", C.c_float), ("ms_d2h", C.c_float),
                ("ms_map_search", C.c_float), ("ms_map_dp", C.c_float), ("ms_map_enqueue", C.c_float)]


TEXT_OK, TEXT_CR, TEXT_RAGGED, TEXT_BAD_RECORD, TEXT_LONG_NAME, TEXT_SAM_CAP, TEXT_TOO_LARGE, TEXT_UNEQUAL, TEXT_INTERNAL, TEXT_DEFERRED = range(10)


class UrmapxError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: {strerror(code)} ({code})")


_lib = None


def lib():
    """Load liburmapx.so (built in-tree by __graft_entry__.build()); fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the HIP library is the only compute path; there is no fallback)")
    L = C.CDLL(LIB_PATH)
    vp, cp, u32, u64, i32 = C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_int
    L.urmapx_params_for_method.argtypes = [C.c_uint, C.POINTER(Params)]
    L.urmapx_index_open.argtypes = [cp, C.POINTER(vp)]
    L.urmapx_index_wrap_host.argtypes = [u32, u32, u64, vp, vp, u32, u32, vp, vp, cp, C.POINTER(vp)]
    L.urmapx_index_wrap_device.argtypes = [i32, u32, u32, u64, vp, vp, u32, u32, vp, vp, cp, C.POINTER(vp)]
    L.urmapx_index_upload.argtypes = [vp, i32]
    L.urmapx_index_replicate.argtypes = [vp, i32, C.POINTER(vp)]
    L.urmapx_index_close.argtypes = [vp]
    L.urmapx_index_close.restype = None
    for name, rt in (("word_length", u32), ("max_ix", u32), ("slot_count", u64), ("seqdata_size", u32),
                     ("seq_count", u32), ("chain_row_bytes", u64)):
        f = getattr(L, "urmapx_index_" + name)
        f.restype = rt
        f.argtypes = [vp]
    L.urmapx_index_validate.argtypes = [vp, C.POINTER(ValidateReport)]
    L.urmapx_checksum_device.argtypes = [i32, vp, u64, C.POINTER(u64)]
    L.urmapx_index_checksum.argtypes = [vp, C.POINTER(u64)]
    L.urmapx_index_layout_checksum.argtypes = [vp, C.POINTER(u64)]
    L.urmapx_index_label.restype = cp
    L.urmapx_index_label.argtypes = [vp, u32]
    L.urmapx_index_seq_length.restype = u32
    L.urmapx_index_seq_length.argtypes = [vp, u32]
    L.urmapx_index_seq_offset.restype = u32
    L.urmapx_index_seq_offset.argtypes = [vp, u32]
    L.urmapx_ctx_create.argtypes = [vp, i32, C.POINTER(Params), C.POINTER(vp)]
    L.urmapx_ctx_destroy.argtypes = [vp]
    L.urmapx_ctx_destroy.restype = None
    L.urmapx_map_se.argtypes = [vp, vp, vp, u32, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.urmapx_map_pe.argtypes = [vp, vp, vp, u32, vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.urmapx_sam_pe.restype = C.c_size_t
    L.urmapx_sam_pe.argtypes = [vp, vp, vp, vp, cp, vp, vp, u32, cp, vp, vp, u32, vp, C.c_size_t]
    L.urmapx_map_se_device.argtypes = [vp, vp, vp, u32, u64, u32, vp, vp, vp]
    L.urmapx_map_pe_device.argtypes = [vp, vp, vp, u32, u64, u32, vp, vp, vp]
    L.urmapx_ctx_set_pe_veryfast.argtypes = [vp, i32]
    L.urmapx_ctx_sync.argtypes = [vp]
    L.urmapx_ctx_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float * 2)]
    L.urmapx_ctx_phase_cycles.argtypes = [vp, C.POINTER(C.c_uint64 * 12)]
    L.urmapx_ctx_read_cycles.argtypes = [vp, vp, u32]
    L.urmapx_seed_probe_device.argtypes = [vp, vp, vp, u32, u64, u32, C.POINTER(C.c_float)]
    L.urmapx_ctx_stage_ms.argtypes = [vp, C.POINTER(C.c_float * 7)]
    L.urmapx_ctx_phase3.argtypes = [vp, C.POINTER(C.c_float * 3), C.POINTER(C.c_uint32 * 2)]
    L.urmapx_ctx_round_ms.argtypes = [vp, C.POINTER(C.c_float * 16), C.POINTER(C.c_int)]
    L.urmapx_ctx_dp_stats.argtypes = [vp, C.POINTER(C.c_uint32 * 8)]
    L.urmapx_seed_probe.argtypes = [vp, vp, vp, u32, vp, vp, vp]
    L.urmapx_viterbi_batch.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp, vp, vp, vp]
    L.urmapx_make_ufi.argtypes = [cp, cp, u32, u32, u64]
    L.urmapx_build_slots.argtypes = [vp, u32, u32, u32, u64, vp, C.POINTER(u32)]
    L.urmapx_make_ufi_gpu.argtypes = [i32, cp, cp, u32, u32, u64]
    L.urmapx_build_slots_gpu.argtypes = [i32, vp, vp, u32, u32, u32, u64, vp, C.POINTER(u32)]
    L.urmapx_sam_se.restype = C.c_size_t
    L.urmapx_sam_se.argtypes = [vp, vp, vp, cp, vp, vp, u32, vp, C.c_size_t]
    L.urmapx_sam_header_sq.restype = C.c_size_t
    L.urmapx_sam_header_sq.argtypes = [vp, vp, C.c_size_t]
    L.urmapx_strerror.restype = cp
    L.urmapx_strerror.argtypes = [i32]
    L.urmapx_device_arch.restype = cp
    L.urmapx_device_arch.argtypes = [vp]
    L.urmapx_ctx_gather_microbench.argtypes = [vp, u64, C.POINTER(C.c_double)]
    L.urmapx_ctx_set_pair_info.argtypes = [vp, i32]
    L.urmapx_ctx_get_pair_info.argtypes = [vp, vp, u32]
    L.urmapx_tab_pe.restype = C.c_size_t
    L.urmapx_tab_pe.argtypes = [vp, vp, vp, vp, cp, u32, u32, i32, vp, C.c_size_t]
    L.urmapx_map_files.argtypes = [vp, C.POINTER(MapOptions), cp, cp, cp, cp, C.POINTER(MapReport), cp, C.c_size_t]
    L.urmapx_host_pool_trim.argtypes = []
    L.urmapx_host_pool_trim.restype = None
    L.urmapx_text_create.argtypes = [vp, C.POINTER(vp)]
    L.urmapx_text_destroy.argtypes = [vp]
    L.urmapx_text_destroy.restype = None
    L.urmapx_text_map_se.argtypes = [vp, vp, C.c_size_t, C.c_uint, vp, C.c_size_t, C.POINTER(TextReport)]
    L.urmapx_text_map_pe.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.c_uint, vp, C.c_size_t, C.POINTER(TextReport)]
    L.urmapx_text_fetch_sam.argtypes = [vp, vp, C.c_size_t, C.POINTER(TextReport)]
    L.urmapx_text_set_deferred.argtypes = [vp, C.c_int]
    L.urmapx_text_wait.argtypes = [vp, C.POINTER(TextReport)]
    L.urmapx_fastq_open.argtypes = [cp, C.POINTER(vp)]
    L.urmapx_fastq_next.restype = C.c_int64
    L.urmapx_fastq_next.argtypes = [vp, u32] + [C.POINTER(vp)] * 5
    L.urmapx_fastq_error.restype = cp
    L.urmapx_fastq_error.argtypes = [vp]
    L.urmapx_fastq_close.restype = None
    L.urmapx_fastq_close.argtypes = [vp]
    _lib = L
    return L


def strerror(code):
    return lib().urmapx_strerror(code).decode()


def _check(rc, what, allow=()):
    if rc != 0 and rc not in allow:
        raise UrmapxError(rc, what)
    return rc


def params_for_method(method=6) -> Params:
    """State1::SetMethod (state1.cpp:147-183): 6 = default, 7 = -veryfast."""
    p = Params()
    _check(lib().urmapx_params_for_method(method, C.byref(p)), "urmapx_params_for_method")
    return p


def make_ufi(fasta, ufi, slots, word_length=24, max_ix=32):
    """urmap -make_ufi FASTA -output UFI -slots N (ufindexio.cpp:117-179)."""
    _check(lib().urmapx_make_ufi(os.fsencode(fasta), os.fsencode(ufi), word_length, max_ix, slots), "urmapx_make_ufi")


def build_slots(seqdata: np.ndarray, slots, word_length=24, max_ix=32) -> np.ndarray:
    """UFIndex::MakeIndex on a concatenated upper-case sequence store -> 5*slots byte slot table."""
    seqdata = np.ascontiguousarray(seqdata, dtype=np.uint8)
    blob = np.empty(5 * slots + 8, dtype=np.uint8)
    blob[5 * slots:] = 0
    trunc = C.c_uint32(0)
    _check(lib().urmapx_build_slots(seqdata.ctypes.data, len(seqdata), word_length, max_ix, slots, blob.ctypes.data,
                                    C.byref(trunc)), "urmapx_build_slots")
    return blob


def make_ufi_gpu(device, fasta, ufi, slots, word_length=24, max_ix=32):
    """-make_ufi with the counting passes, head slots and overflow list made on the GPU; byte-identical output."""
    _check(lib().urmapx_make_ufi_gpu(device, os.fsencode(fasta), os.fsencode(ufi), word_length, max_ix, slots), "urmapx_make_ufi_gpu")


def build_slots_gpu(device, slots, seqdata: np.ndarray | None = None, d_seq_ptr=None, size=None, word_length=24, max_ix=32) -> np.ndarray:
    """UFIndex::MakeIndex -> 5*slots byte slot table; the sequence store comes from a host array or is resident on
    `device` already (d_seq_ptr, size)."""
    blob = np.empty(5 * slots + 8, dtype=np.uint8)
    blob[5 * slots:] = 0
    trunc = C.c_uint32(0)
    if seqdata is not None:
        seqdata = np.ascontiguousarray(seqdata, dtype=np.uint8)
        size = len(seqdata)
    _check(lib().urmapx_build_slots_gpu(device, seqdata.ctypes.data if seqdata is not None else None, d_seq_ptr, size, word_length,
                                        max_ix, slots, blob.ctypes.data, C.byref(trunc)), "urmapx_build_slots_gpu")
    return blob


def checksum_device(device, d_ptr, nbytes):
    """urmapx_checksum_device: the checksum of `nbytes` device-resident bytes at the 8-byte-aligned address d_ptr"""
    out = C.c_uint64(0)
    _check(lib().urmapx_checksum_device(device, d_ptr, nbytes, C.byref(out)), "urmapx_checksum_device")
    return int(out.value)


class Index:
    """UFIndex (read side): parsed .ufi + its copy in HBM."""

    def __init__(self, handle, keep=()):
        self.h = C.c_void_p(handle)
        self._keep = keep

    @classmethod
    def open(cls, path):
        h = C.c_void_p()
        _check(lib().urmapx_index_open(os.fsencode(path), C.byref(h)), f"urmapx_index_open({path})")
        return cls(h.value)

    @classmethod
    def open_device(cls, path, device=0):
        """UFIndex::FromFile straight into the HBM of `device` (no host copy of the arrays is kept): what the command line loads with."""
        h = C.c_void_p()
        _check(lib().urmapx_index_open_device(os.fsencode(path), int(device), C.byref(h)), f"urmapx_index_open_device({path})")
        return cls(h.value)

    @classmethod
    def wrap_host(cls, word_length, max_ix, slot_count, blob, seqdata, seq_lengths, offsets, labels):
        sl = np.ascontiguousarray(seq_lengths, dtype=np.uint32)
        of = np.ascontiguousarray(offsets, dtype=np.uint32)
        lab = b"".join(l.encode() + b"\0" for l in labels)
        h = C.c_void_p()
        _check(lib().urmapx_index_wrap_host(word_length, max_ix, slot_count, blob.ctypes.data, seqdata.ctypes.data,
                                            len(seqdata), len(labels), sl.ctypes.data, of.ctypes.data, lab,
                                            C.byref(h)), "urmapx_index_wrap_host")
        return cls(h.value, keep=(blob, seqdata, sl, of, lab))

    @classmethod
    def wrap_device(cls, device, word_length, max_ix, slot_count, d_blob_ptr, d_seq_ptr, seqdata_size, seq_lengths,
                    offsets, labels, keep=()):
        """Adopt arrays already resident in HBM (raw device pointers, e.g. torch tensor .data_ptr())."""
        sl = np.ascontiguousarray(seq_lengths, dtype=np.uint32)
        of = np.ascontiguousarray(offsets, dtype=np.uint32)
        lab = b"".join(l.encode() + b"\0" for l in labels)
        h = C.c_void_p()
        _check(lib().urmapx_index_wrap_device(device, word_length, max_ix, slot_count, d_blob_ptr, d_seq_ptr,
                                              seqdata_size, len(labels), sl.ctypes.data, of.ctypes.data, lab,
                                              C.byref(h)), "urmapx_index_wrap_device")
        return cls(h.value, keep=tuple(keep) + (sl, of, lab))

    def upload(self, device=0):
        _check(lib().urmapx_index_upload(self.h, device), "urmapx_index_upload")
        return self

    def replicate(self, device):
        """Another replica of this index in the HBM of `device` (one per GPU; reads are sharded across them)."""
        h = C.c_void_p()
        _check(lib().urmapx_index_replicate(self.h, device, C.byref(h)), "urmapx_index_replicate")
        return Index(h.value, keep=(self,))

    def chain_row_bytes(self): return int(lib().urmapx_index_chain_row_bytes(self.h))

    def validate(self):
        """UFIndex::Validate (ufindex.cpp:611-658) as one device pass over the resident table -> (ok, report dict)."""
        r = ValidateReport()
        rc = lib().urmapx_index_validate(self.h, C.byref(r))
        if rc not in (0, E_FORMAT):
            raise UrmapxError(rc, "urmapx_index_validate")
        return rc == 0, {n: getattr(r, n) for n, _ in ValidateReport._fields_}

    def checksum(self):
        """(slot table, sequence store): urmapx_index_checksum over the resident arrays -- which table and which genome this is"""
        out = (C.c_uint64 * 2)()
        _check(lib().urmapx_index_checksum(self.h, out), "urmapx_index_checksum")
        return int(out[0]), int(out[1])

    def layout_checksum(self):
        """(slot16, chain rows): urmapx_index_layout_checksum -- the layouts derived from the table at upload (0: not built)"""
        out = (C.c_uint64 * 2)()
        _check(lib().urmapx_index_layout_checksum(self.h, out), "urmapx_index_layout_checksum")
        return int(out[0]), int(out[1])

    @property
    def word_length(self): return lib().urmapx_index_word_length(self.h)
    @property
    def max_ix(self): return lib().urmapx_index_max_ix(self.h)
    @property
    def slot_count(self): return lib().urmapx_index_slot_count(self.h)
    @property
    def seqdata_size(self): return lib().urmapx_index_seqdata_size(self.h)

    def directory(self):
        n = lib().urmapx_index_seq_count(self.h)
        return [(lib().urmapx_index_label(self.h, i).decode(), lib().urmapx_index_seq_length(self.h, i),
                 lib().urmapx_index_seq_offset(self.h, i)) for i in range(n)]

    def sam_header_sq(self) -> bytes:
        buf = C.create_string_buffer(1 << 20)
        n = lib().urmapx_sam_header_sq(self.h, buf, len(buf))
        return buf.raw[:n]

    def sam_se(self, results: np.ndarray, ops: np.ndarray, labels, bases: np.ndarray, offs: np.ndarray,
               quals: np.ndarray) -> bytes:
        """SAM records (State1::Output1 -> SetSAM) of a mapped batch, in input order.  Host-side text only."""
        results = np.ascontiguousarray(results)
        ops = np.ascontiguousarray(ops, dtype=np.uint16)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        out = []
        buf = C.create_string_buffer(1 << 16)
        ops_ptr = ops.ctypes.data if len(ops) else None
        for i in range(len(results)):
            o, e = int(offs[i]), int(offs[i + 1])
            n = lib().urmapx_sam_se(self.h, results[i:i + 1].ctypes.data, ops_ptr, labels[i].encode(),
                                    bases[o:e].ctypes.data, quals[o:e].ctypes.data, e - o, buf, len(buf))
            if n == 0:
                raise UrmapxError(-5, "urmapx_sam_se: record does not fit")
            out.append(buf.raw[:n])
        return b"".join(out)

    def tab_pe(self, res, info, labels, offs, sam_on=True):
        """-tabbedout lines (State2::OutputTab2) for pairs interleaved as in map_pe; labels = per read."""
        L = lib()
        out = []
        buf = C.create_string_buffer(4096)
        res = np.ascontiguousarray(res)
        info = np.ascontiguousarray(info)
        for i in range(len(info)):
            l1 = int(offs[2 * i + 1] - offs[2 * i]); l2 = int(offs[2 * i + 2] - offs[2 * i + 1])
            k = L.urmapx_tab_pe(self.h, res[2 * i:].ctypes.data, res[2 * i + 1:].ctypes.data, info[i:].ctypes.data,
                                labels[2 * i].encode("latin-1"), l1, l2, int(sam_on), buf, len(buf))
            out.append(buf.raw[:k])
        return b"".join(out)

    def sam_pe(self, results: np.ndarray, ops: np.ndarray, labels, bases: np.ndarray, offs: np.ndarray,
               quals: np.ndarray) -> bytes:
        """SAM records of mapped PAIRS (reads 2i, 2i+1 = mates of pair i): State2::SetSAM2 + SetSAM."""
        results = np.ascontiguousarray(results)
        ops = np.ascontiguousarray(ops, dtype=np.uint16)
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        quals = np.ascontiguousarray(quals, dtype=np.uint8)
        out = []
        buf = C.create_string_buffer(1 << 16)
        ops_ptr = ops.ctypes.data if len(ops) else None
        for i in range(0, len(results), 2):
            o1, e1, e2 = int(offs[i]), int(offs[i + 1]), int(offs[i + 2])
            n = lib().urmapx_sam_pe(self.h, results[i:i + 1].ctypes.data, results[i + 1:i + 2].ctypes.data, ops_ptr,
                                    labels[i].encode(), bases[o1:e1].ctypes.data, quals[o1:e1].ctypes.data, e1 - o1,
                                    labels[i + 1].encode(), bases[e1:e2].ctypes.data, quals[e1:e2].ctypes.data, e2 - e1,
                                    buf, len(buf))
            if n == 0:
                raise UrmapxError(-5, "urmapx_sam_pe: record does not fit")
            out.append(buf.raw[:n])
        return b"".join(out)

    def close(self):
        if self.h:
            lib().urmapx_index_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gunzip_file(gz_path, out_path, threads=0):
    """the .gz reader of urmapx_map_files alone (pgzip.h) -> (bytes written, by the parallel road, through zlib)"""
    L = lib()
    L.urmapx_gunzip_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64 * 3)]
    st = (C.c_uint64 * 3)()
    _check(L.urmapx_gunzip_file(os.fsencode(gz_path), os.fsencode(out_path), threads, C.byref(st)), f"urmapx_gunzip_file({gz_path})")
    return int(st[0]), int(st[1]), int(st[2])


def map_files(index: "Index", fastq1, fastq2=None, samout=None, tabout=None, first_gpu=0, gpus=1, streams=2, host_threads=0,
              batch=0, veryfast=False, minq=10, cmdline=None, allow_unsupported=False, sam_shards=0, discard_sam=False):
    """urmap -map / -map2 file to file (cmd_map / cmd_map2) on an index that has its host arrays or is resident on
    first_gpu.  -> dict of State1::HitStats' counters and stage times."""
    o = MapOptions(first_gpu, gpus, streams, host_threads, batch, int(veryfast), minq, cmdline.encode() if cmdline else None,
                   int(sam_shards), int(discard_sam))
    rep = MapReport()
    err = C.create_string_buffer(1024)
    enc = lambda p: os.fsencode(p) if p else None
    rc = lib().urmapx_map_files(index.h, C.byref(o), enc(fastq1), enc(fastq2), enc(samout), enc(tabout), C.byref(rep), err, len(err))
    if rc != 0 and not (rc == E_UNSUPPORTED and allow_unsupported):
        raise UrmapxError(rc, "urmapx_map_files: " + err.value.decode("latin-1"))
    return {k: getattr(rep, k) for k, _ in MapReport._fields_}


def decode_path(ops: np.ndarray) -> str:
    """Run-length path arena entries -> the reference's M/D/I path string."""
    return "".join("MDI"[int(o) & 3] * (int(o) >> 2) for o in ops)


class Mapper:
    """One mapping context (the State1 of one OMP thread in map.cpp:11-25), batch form."""

    def __init__(self, index: Index, device=0, method=6, params: Params | None = None):
        self.index = index
        self.device = device
        self.params = params if params is not None else params_for_method(method)
        h = C.c_void_p()
        self._text = None
        self.h = None
        _check(lib().urmapx_ctx_create(index.h, device, C.byref(self.params), C.byref(h)), "urmapx_ctx_create")
        self.h = h

    @property
    def arch(self):
        return lib().urmapx_device_arch(self.h).decode()

    def map_se(self, bases: np.ndarray, offs: np.ndarray, allow_unsupported=False):
        """-> (results structured array, path op arena).  Raises UrmapxError(E_UNSUPPORTED) if any read
        fell outside the device domain unless allow_unsupported (then check results['status'])."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        res = np.zeros(n, dtype=RESULT_DTYPE)
        cap = max(1, n * MAX_PATH_OPS)
        ops = np.zeros(cap, dtype=np.uint16)
        used = C.c_size_t(0)
        rc = lib().urmapx_map_se(self.h, bases.ctypes.data, offs.ctypes.data, n, res.ctypes.data, ops.ctypes.data, cap,
                                 C.byref(used))
        _check(rc, "urmapx_map_se", allow=(E_UNSUPPORTED,) if allow_unsupported else ())
        return res, ops[: used.value].copy()

    def set_pe_veryfast(self, on=True):
        _check(lib().urmapx_ctx_set_pe_veryfast(self.h, int(on)), "urmapx_ctx_set_pe_veryfast")

    def map_pe(self, bases: np.ndarray, offs: np.ndarray, allow_unsupported=False):
        """Pairs interleaved: reads 2i, 2i+1 are R1, R2 of pair i.  -> (results[2*npairs], path op arena)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        assert n % 2 == 0
        res = np.zeros(n, dtype=RESULT_DTYPE)
        cap = max(1, n * MAX_PATH_OPS)
        ops = np.zeros(cap, dtype=np.uint16)
        used = C.c_size_t(0)
        rc = lib().urmapx_map_pe(self.h, bases.ctypes.data, offs.ctypes.data, n // 2, res.ctypes.data, ops.ctypes.data,
                                 cap, C.byref(used))
        _check(rc, "urmapx_map_pe", allow=(E_UNSUPPORTED,) if allow_unsupported else ())
        return res, ops[: used.value].copy()

    def map_se_device(self, d_bases_ptr, d_offs_ptr, n, total_bases, max_read_len, d_results_ptr, d_path_ops_ptr,
                      d_path_used_ptr):
        _check(lib().urmapx_map_se_device(self.h, d_bases_ptr, d_offs_ptr, n, total_bases, max_read_len,
                                          d_results_ptr, d_path_ops_ptr, d_path_used_ptr), "urmapx_map_se_device")

    def map_pe_device(self, d_bases_ptr, d_offs_ptr, npairs, total_bases, max_read_len, d_results_ptr, d_path_ops_ptr,
                      d_path_used_ptr):
        _check(lib().urmapx_map_pe_device(self.h, d_bases_ptr, d_offs_ptr, npairs, total_bases, max_read_len,
                                          d_results_ptr, d_path_ops_ptr, d_path_used_ptr), "urmapx_map_pe_device")

    def sync(self):
        _check(lib().urmapx_ctx_sync(self.h), "urmapx_ctx_sync")

    def last_kernel_ms(self):
        ms = (C.c_float * 2)()
        _check(lib().urmapx_ctx_last_kernel_ms(self.h, C.byref(ms)), "urmapx_ctx_last_kernel_ms")
        return float(ms[0]), float(ms[1])

    def stage_ms(self):
        """ms of the launches of the last single-end device call: search, its DP launches (summed), its finalize launches
        (summed), the same three for the second pass, the general kernel over what both left flagged."""
        ms = (C.c_float * 7)()
        _check(lib().urmapx_ctx_stage_ms(self.h, C.byref(ms)), "urmapx_ctx_stage_ms")
        return [float(x) for x in ms]

    def seed_probe_device(self, d_bases_ptr, d_offs_ptr, n, total_bases, max_read_len):
        """the probe launch alone over reads resident in HBM -> its ms on the context's stream"""
        ms = C.c_float(0)
        _check(lib().urmapx_seed_probe_device(self.h, d_bases_ptr, d_offs_ptr, n, total_bases, max_read_len, C.byref(ms)), "urmapx_seed_probe_device")
        return float(ms.value)

    def phase3(self):
        """The search stage of the last single-end device call when phase 3 is parked (round 5): ms of the first search launch, of
        phase 3's DP launch and of the launch over the reads parked at phase 3; DpJobs made for phase 3, reads parked there."""
        ms = (C.c_float * 3)()
        st = (C.c_uint32 * 2)()
        _check(lib().urmapx_ctx_phase3(self.h, C.byref(ms), C.byref(st)), "urmapx_ctx_phase3")
        return [float(x) for x in ms], [int(x) for x in st]

    def round_ms(self):
        """ms of the first pass's phase-6 launches one by one: [(dp, finalize) per round]"""
        ms = (C.c_float * 16)()
        n = C.c_int(0)
        _check(lib().urmapx_ctx_round_ms(self.h, C.byref(ms), C.byref(n)), "urmapx_ctx_round_ms")
        return [(float(ms[2 * r]), float(ms[2 * r + 1])) for r in range(n.value)]

    def dp_rounds(self):
        """the rounds of phase 6 in the last single-end call as text: 'HSPs [0,2), [2,16), [16,...) of a read'"""
        lo = (C.c_uint32 * 8)()
        n = C.c_int(0)
        L = lib()
        L.urmapx_ctx_dp_rounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint32 * 8), C.POINTER(C.c_int)]
        _check(L.urmapx_ctx_dp_rounds(self.h, C.byref(lo), C.byref(n)), "urmapx_ctx_dp_rounds")
        parts = [f"[{lo[r]},{lo[r + 1]})" if r + 1 < n.value else f"[{lo[r]},...)" for r in range(n.value)]
        return "HSPs " + ", ".join(parts) + " of a read"

    def dp_stats(self):
        """per pass: (HSPs handed to the DP launches, reads parked, DPs the ordered replay looked at, jobs gated before their DP)"""
        out = (C.c_uint32 * 8)()
        _check(lib().urmapx_ctx_dp_stats(self.h, C.byref(out)), "urmapx_ctx_dp_stats")
        return [int(x) for x in out]

    def set_pair_info(self, on=True):
        _check(lib().urmapx_ctx_set_pair_info(self.h, int(on)), "urmapx_ctx_set_pair_info")

    def pair_info(self, npairs):
        """urmapx_pair_info records of the last paired-end call (needs set_pair_info(True) before it)."""
        out = np.zeros(npairs, dtype=PAIR_INFO_DTYPE)
        _check(lib().urmapx_ctx_get_pair_info(self.h, out.ctypes.data, npairs), "urmapx_ctx_get_pair_info")
        return out

    def gather_microbench(self, n_loads=1 << 28):
        """Random 5-byte slot reads per second over the resident slot table (measurement aid)."""
        r = C.c_double(0.0)
        _check(lib().urmapx_ctx_gather_microbench(self.h, n_loads, C.byref(r)), "urmapx_ctx_gather_microbench")
        return r.value

    def phase_cycles(self):
        out = (C.c_uint64 * 12)()
        _check(lib().urmapx_ctx_phase_cycles(self.h, C.byref(out)), "urmapx_ctx_phase_cycles")
        return [int(x) for x in out]

    def read_cycles(self, n):
        """Shader cycles each of the first n reads of the last single-end call took (URMAPX_PHASE_STATS=1)."""
        out = np.zeros(n, dtype=np.uint32)
        _check(lib().urmapx_ctx_read_cycles(self.h, out.ctypes.data, n), "urmapx_ctx_read_cycles")
        return out.astype(np.int64) * 16

    def seed_probe(self, bases: np.ndarray, offs: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offs = np.ascontiguousarray(offs, dtype=np.uint64)
        n = len(offs) - 1
        total = int(offs[-1])
        slots = np.zeros(2 * total, dtype=np.uint64)
        tallies = np.zeros(2 * total, dtype=np.uint8)
        positions = np.zeros(2 * total, dtype=np.uint32)
        _check(lib().urmapx_seed_probe(self.h, bases.ctypes.data, offs.ctypes.data, n, slots.ctypes.data,
                                       tallies.ctypes.data, positions.ctypes.data), "urmapx_seed_probe")
        return slots, tallies, positions

    def viterbi_batch(self, pairs, flags):
        """pairs: list of (A bytes, B bytes); flags: list of ints (bit0 Left, bit1 Right)."""
        n = len(pairs)
        a = np.frombuffer(b"".join(p[0] for p in pairs), dtype=np.uint8)
        b = np.frombuffer(b"".join(p[1] for p in pairs), dtype=np.uint8)
        ao = np.zeros(n + 1, dtype=np.uint32)
        bo = np.zeros(n + 1, dtype=np.uint32)
        ao[1:] = np.cumsum([len(p[0]) for p in pairs])
        bo[1:] = np.cumsum([len(p[1]) for p in pairs])
        fl = np.ascontiguousarray(flags, dtype=np.uint8)
        scores = np.zeros(n, dtype=np.float32)
        status = np.zeros(n, dtype=np.uint8)
        ops = np.zeros(n * MAX_PATH_OPS, dtype=np.uint16)
        nops = np.zeros(n, dtype=np.uint16)
        a_ = np.ascontiguousarray(a) if len(a) else np.zeros(1, np.uint8)
        b_ = np.ascontiguousarray(b) if len(b) else np.zeros(1, np.uint8)
        _check(lib().urmapx_viterbi_batch(self.h, a_.ctypes.data, ao.ctypes.data, b_.ctypes.data, bo.ctypes.data,
                                          fl.ctypes.data, n, scores.ctypes.data, status.ctypes.data, ops.ctypes.data,
                                          nops.ctypes.data), "urmapx_viterbi_batch")
        paths = [decode_path(ops[i * MAX_PATH_OPS: i * MAX_PATH_OPS + int(nops[i])]) for i in range(n)]
        return scores, status, paths

    def map_text_se(self, fastq: bytes, minq=10, sam_cap=None):
        """A chunk of FASTQ text (cut after a record's last newline) -> (the SAM text of its records | None, report dict).
        None with report['reason'] != TEXT_OK: the device parser does not take the chunk as it is (see urmapx.h)."""
        if self._text is None:
            t = C.c_void_p()
            _check(lib().urmapx_text_create(self.h, C.byref(t)), "urmapx_text_create")
            self._text = t
        src = np.frombuffer(fastq, dtype=np.uint8)
        cap = sam_cap if sam_cap is not None else 2 * len(src) + 4096 * 64
        out = np.empty(max(1, cap), dtype=np.uint8)
        rep = TextReport()
        _check(lib().urmapx_text_map_se(self._text, src.ctypes.data if len(src) else None, len(src), minq, out.ctypes.data, cap, C.byref(rep)),
               "urmapx_text_map_se")
        d = {k: int(getattr(rep, k)) for k, _ in TextReport._fields_}
        if rep.reason != TEXT_OK:
            return None, d
        return out[: rep.sam_bytes].tobytes(), d

    def map_text_pe(self, fastq1: bytes, fastq2: bytes, minq=10, sam_cap=None):
        """Chunks of the two mate files with the same number of records -> (SAM text of the pairs | None, report dict)."""
        if self._text is None:
            t = C.c_void_p()
            _check(lib().urmapx_text_create(self.h, C.byref(t)), "urmapx_text_create")
            self._text = t
        a, b = np.frombuffer(fastq1, dtype=np.uint8), np.frombuffer(fastq2, dtype=np.uint8)
        cap = sam_cap if sam_cap is not None else 2 * (len(a) + len(b)) + 4096 * 64
        out = np.empty(max(1, cap), dtype=np.uint8)
        rep = TextReport()
        _check(lib().urmapx_text_map_pe(self._text, a.ctypes.data if len(a) else out.ctypes.data, len(a), b.ctypes.data if len(b) else out.ctypes.data,
                                        len(b), minq, out.ctypes.data, cap, C.byref(rep)), "urmapx_text_map_pe")
        d = {k: int(getattr(rep, k)) for k, _ in TextReport._fields_}
        if rep.reason != TEXT_OK:
            return None, d
        return out[: rep.sam_bytes].tobytes(), d

    def map_text_se_stream(self, chunks, minq=10, sam_caps=None):
        """Chunks of one FASTQ file through the context with the copy back deferred (urmapx_text_set_deferred): chunk i + 1 is handed
        over before chunk i's text is waited for, as a lane of urmapx_map_files does.  -> [(SAM text | None, report dict)] per chunk.
        sam_caps[i] (optional): the buffer offered for chunk i; too small -> urmapx_text_fetch_sam into a larger one, deferred too."""
        if self._text is None:
            t = C.c_void_p()
            _check(lib().urmapx_text_create(self.h, C.byref(t)), "urmapx_text_create")
            self._text = t
        _check(lib().urmapx_text_set_deferred(self._text, 1), "urmapx_text_set_deferred")
        res, prev = [], None

        def finish(item):
            out, d0 = item
            if d0["reason"] != TEXT_DEFERRED:
                return (b"" if d0["reason"] == TEXT_OK else None), d0
            rep = TextReport()
            _check(lib().urmapx_text_wait(self._text, C.byref(rep)), "urmapx_text_wait")
            d = {k: (int(getattr(rep, k)) if not k.startswith("ms_") else float(getattr(rep, k))) for k, _ in TextReport._fields_}
            return (out[: rep.sam_bytes].tobytes() if rep.reason == TEXT_OK else None), d

        try:
            for i, fastq in enumerate(chunks):
                src = np.frombuffer(fastq, dtype=np.uint8)
                cap = sam_caps[i] if sam_caps is not None and sam_caps[i] is not None else 2 * len(src) + 4096 * 64
                out = np.empty(max(1, cap), dtype=np.uint8)
                rep = TextReport()
                _check(lib().urmapx_text_map_se(self._text, src.ctypes.data if len(src) else None, len(src), minq, out.ctypes.data, cap, C.byref(rep)),
                       "urmapx_text_map_se")
                if rep.reason == TEXT_SAM_CAP:
                    out = np.empty(int(rep.sam_bytes) + 64, dtype=np.uint8)
                    _check(lib().urmapx_text_fetch_sam(self._text, out.ctypes.data, len(out), C.byref(rep)), "urmapx_text_fetch_sam")
                d = {k: (int(getattr(rep, k)) if not k.startswith("ms_") else float(getattr(rep, k))) for k, _ in TextReport._fields_}
                if prev is not None:
                    item, prev = prev, None
                    res.append(finish(item))
                prev = (out, d)
            if prev is not None:
                item, prev = prev, None
                res.append(finish(item))
        finally:
            #  after an exception a chunk may still be on its way: it is waited for before the mode is switched back, and nothing
            # is raised from here -- the exception that ended the loop is the one the caller sees
            rep = TextReport()
            while lib().urmapx_text_wait(self._text, C.byref(rep)) == 0:
                pass
            lib().urmapx_text_set_deferred(self._text, 0)
        return res

    def fetch_text_sam(self, sam_cap):
        """After map_text_se returned TEXT_SAM_CAP: the text of that chunk (the search is not run again)."""
        out = np.empty(max(1, sam_cap), dtype=np.uint8)
        rep = TextReport()
        _check(lib().urmapx_text_fetch_sam(self._text, out.ctypes.data, sam_cap, C.byref(rep)), "urmapx_text_fetch_sam")
        d = {k: int(getattr(rep, k)) for k, _ in TextReport._fields_}
        return (out[: rep.sam_bytes].tobytes() if rep.reason == TEXT_OK else None), d

    def close(self):
        if self._text is not None:
            lib().urmapx_text_destroy(self._text)
            self._text = None
        if self.h:
            lib().urmapx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def interleave_pairs(a, b):
    """(labels, bases, offs, quals) of R1 and of R2 -> one interleaved set (R1_0, R2_0, R1_1, ...)."""
    la, ba, oa, qa = a
    lb, bb, ob, qb = b
    assert len(la) == len(lb)
    labels, seqs, quals = [], [], []
    for i in range(len(la)):
        labels += [la[i], lb[i]]
        seqs += [ba[int(oa[i]):int(oa[i + 1])], bb[int(ob[i]):int(ob[i + 1])]]
        quals += [qa[int(oa[i]):int(oa[i + 1])], qb[int(ob[i]):int(ob[i + 1])]]
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return labels, np.concatenate(seqs), offs, np.concatenate(quals)


class FastqReader:
    """ctypes view of urmapx_fastq_* (the product's FASTQSeqSource): iterate batches of records."""

    def __init__(self, path):
        self._h = C.c_void_p()
        rc = lib().urmapx_fastq_open(os.fsencode(path), C.byref(self._h))
        if rc:
            raise UrmapxError(rc, f"urmapx_fastq_open({path})")

    def next(self, max_reads):
        """-> (labels list[str], bases uint8, offs uint64, quals uint8) or None at end of file."""
        L = lib()
        pb, pq, po, pl, plo = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = L.urmapx_fastq_next(self._h, max_reads, C.byref(pb), C.byref(pq), C.byref(po), C.byref(pl), C.byref(plo))
        if n < 0:
            raise ValueError(L.urmapx_fastq_error(self._h).decode())
        if n == 0:
            return None
        offs = np.ctypeslib.as_array(C.cast(po, C.POINTER(C.c_uint64)), shape=(n + 1,)).copy()
        nb = int(offs[-1])
        if nb:
            bases = np.ctypeslib.as_array(C.cast(pb, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
            quals = np.ctypeslib.as_array(C.cast(pq, C.POINTER(C.c_uint8)), shape=(nb,)).copy()
        else:
            bases = np.zeros(0, dtype=np.uint8)
            quals = np.zeros(0, dtype=np.uint8)
        lo = np.ctypeslib.as_array(C.cast(plo, C.POINTER(C.c_uint64)), shape=(n,))
        labels = [C.string_at(pl.value + int(o)).decode("latin-1") for o in lo]
        return labels, bases, offs, quals

    def close(self):
        if self._h:
            lib().urmapx_fastq_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_fastq_arrays(path, batch=1 << 16):
    """FASTQ -> (labels, bases uint8, offs uint64, quals uint8) through the product's reader."""
    rd = FastqReader(path)
    labels, bl, ql, lens = [], [], [], []
    while True:
        b = rd.next(batch)
        if b is None:
            break
        labels += b[0]
        bl.append(b[1])
        ql.append(b[3])
        lens.append(np.diff(b[2]))
    rd.close()
    offs = np.zeros(len(labels) + 1, dtype=np.uint64)
    if lens:
        offs[1:] = np.cumsum(np.concatenate(lens))
    bases = np.concatenate(bl) if bl else np.zeros(0, dtype=np.uint8)
    qual = np.concatenate(ql) if ql else np.zeros(0, dtype=np.uint8)
    return labels, bases, offs, qual
