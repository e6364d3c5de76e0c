"""ctypes binding of the parity oracle (oracle/biod_oracle.c).  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "_build", "liboracle.so")

ORC_OK, ORC_EINVAL, ORC_ETRUNC, ORC_ECIGAR, ORC_ENOMEM, ORC_EMD, ORC_EUNSORTED, ORC_ESEQ = 0, -1, -2, -3, -5, -7, -8, -9
MODE_PILEUP_COLUMNS, MODE_MAKE_PILEUP = 0, 1
U64_MAX = (1 << 64) - 1


def build() -> str:
    src = os.path.join(_ROOT, "oracle", "biod_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(src[:-2] + ".h")):
        subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return _SO


class _Fields(C.Structure):
    _fields_ = [("ref_id", C.c_int32), ("pos", C.c_int32), ("bin_mq_nl", C.c_uint32), ("flag_nc", C.c_uint32),
                ("l_seq", C.c_int32), ("next_ref_id", C.c_int32), ("next_pos", C.c_int32), ("tlen", C.c_int32),
                ("bin", C.c_uint32), ("mapq", C.c_uint32), ("l_read_name", C.c_uint32), ("flag", C.c_uint32),
                ("n_cigar_op", C.c_uint32), ("block_size", C.c_uint32),
                ("name_off", C.c_uint64), ("cigar_off", C.c_uint64), ("seq_off", C.c_uint64),
                ("qual_off", C.c_uint64), ("tags_off", C.c_uint64),
                ("bases_covered", C.c_int32), ("end_pos", C.c_int32)]


class _Entry(C.Structure):
    _fields_ = [("read_ordinal", C.c_uint64), ("cigar_op_index", C.c_uint32), ("cigar_op_offset", C.c_uint32),
                ("query_offset", C.c_uint32), ("cigar_op", C.c_uint32), ("base", C.c_uint8), ("qual", C.c_uint8)]


ENTRY_DTYPE = np.dtype([("read_ordinal", "<u8"), ("cigar_op_index", "<u4"), ("cigar_op_offset", "<u4"),
                        ("query_offset", "<u4"), ("cigar_op", "<u4"), ("base", "u1"), ("qual", "u1"), ("_pad", "V6")])
assert ENTRY_DTYPE.itemsize == C.sizeof(_Entry)


class _PileupResult(C.Structure):
    _fields_ = [("status", C.c_int), ("n_columns", C.c_uint64),
                ("col_ref", C.POINTER(C.c_int32)), ("col_pos", C.POINTER(C.c_uint64)),
                ("col_cov", C.POINTER(C.c_uint32)), ("col_nstart", C.POINTER(C.c_uint32)),
                ("col_refbase", C.POINTER(C.c_uint8)), ("col_counts", C.POINTER(C.c_uint32)),
                ("col_entry_off", C.POINTER(C.c_uint64)), ("entries", C.POINTER(_Entry)),
                ("n_entries", C.c_uint64), ("sum_coverage", C.c_uint64), ("max_coverage", C.c_uint32),
                ("pileup_ref_id", C.c_int32)]


class _BenchResult(C.Structure):
    _fields_ = [("n_columns", C.c_uint64), ("sum_coverage", C.c_uint64), ("n_reads", C.c_uint64),
                ("checksum", C.c_uint64), ("max_coverage", C.c_uint32), ("status", C.c_int)]


class _MdOp(C.Structure):
    _fields_ = [("type", C.c_uint8), ("match", C.c_uint32), ("mismatch", C.c_char), ("del_off", C.c_uint32), ("del_len", C.c_uint32)]


class _Soa(C.Structure):
    _fields_ = [("ref_id", C.c_void_p), ("pos", C.c_void_p), ("end_pos", C.c_void_p), ("bin_mq_nl", C.c_void_p),
                ("flag_nc", C.c_void_p), ("l_seq", C.c_void_p), ("seq_off", C.c_void_p), ("cigar_off", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("cigar", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_frame.restype = C.c_int64
        L.orc_frame.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64]
        L.orc_fields_at.restype = None
        L.orc_fields_at.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(_Fields)]
        L.orc_decode_soa.restype = C.c_int
        L.orc_decode_soa.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(_Soa)]
        L.orc_pileup.restype = C.c_int
        L.orc_pileup.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(_PileupResult)]
        L.orc_pileup_chunked.restype = C.c_int
        L.orc_pileup_chunked.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.POINTER(_PileupResult)]
        L.orc_pileup_free.argtypes = [C.POINTER(_PileupResult)]
        L.orc_pileup_bench.restype = C.c_int
        L.orc_pileup_bench.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_uint64, C.POINTER(_BenchResult)]
        L.orc_md_operations.restype = C.c_int64
        L.orc_md_operations.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_MdOp), C.c_size_t]
        L.orc_dna_read.restype = C.c_int64
        L.orc_dna_read.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.orc_dna_reads.restype = C.c_int64
        L.orc_dna_reads.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_size_t]
        L.orc_find_tag.restype = C.c_int
        L.orc_find_tag.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.POINTER(C.c_uint64), C.c_char_p]
        L.orc_cigar_type.restype = C.c_char
        L.orc_cigar_type.argtypes = [C.c_uint32]
        for nm in ("orc_cigar_length", "orc_cigar_is_query_consuming", "orc_cigar_is_reference_consuming",
                   "orc_cigar_is_match_or_mismatch", "orc_cigar_is_clipping"):
            getattr(L, nm).restype = C.c_uint32 if nm == "orc_cigar_length" else C.c_int
            getattr(L, nm).argtypes = [C.c_uint32]
        L.orc_base_code2char.restype = C.c_char
        L.orc_base_code2char.argtypes = [C.c_uint32]
        L.orc_base_char2code.restype = C.c_uint32
        L.orc_base_char2code.argtypes = [C.c_uint8]
        _lib = L
    return _lib


def frame(stream: bytes) -> np.ndarray:
    n = lib().orc_frame(stream, len(stream), None, 0)
    if n < 0:
        raise OracleError(int(n))
    out = np.zeros(n, dtype=np.uint64)
    lib().orc_frame(stream, len(stream), out.ctypes.data, n)
    return out


def fields_at(stream: bytes, off: int) -> _Fields:
    f = _Fields()
    lib().orc_fields_at(stream, off, C.byref(f))
    return f


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle status {code}")
        self.code = code


@dataclass
class Soa:
    rec_off: np.ndarray
    ref_id: np.ndarray
    pos: np.ndarray
    end_pos: np.ndarray
    bin_mq_nl: np.ndarray
    flag_nc: np.ndarray
    l_seq: np.ndarray
    seq_off: np.ndarray
    cigar_off: np.ndarray
    seq: np.ndarray
    qual: np.ndarray
    cigar: np.ndarray


def decode_soa(stream: bytes) -> Soa:
    rec_off = frame(stream)
    n = len(rec_off)
    a = dict(ref_id=np.zeros(n, np.int32), pos=np.zeros(n, np.int32), end_pos=np.zeros(n, np.int32),
             bin_mq_nl=np.zeros(n, np.uint32), flag_nc=np.zeros(n, np.uint32), l_seq=np.zeros(n, np.int32),
             seq_off=np.zeros(n + 1, np.uint64), cigar_off=np.zeros(n + 1, np.uint64))
    s = _Soa(**{k: v.ctypes.data for k, v in a.items()})
    rc = lib().orc_decode_soa(stream, rec_off.ctypes.data, n, C.byref(s))
    if rc:
        raise OracleError(rc)
    seq = np.zeros(int(a["seq_off"][n]), np.uint8)
    qual = np.zeros(int(a["seq_off"][n]), np.uint8)
    cigar = np.zeros(int(a["cigar_off"][n]), np.uint32)
    s.seq, s.qual, s.cigar = seq.ctypes.data, qual.ctypes.data, cigar.ctypes.data
    rc = lib().orc_decode_soa(stream, rec_off.ctypes.data, n, C.byref(s))
    if rc:
        raise OracleError(rc)
    return Soa(rec_off=rec_off, seq=seq, qual=qual, cigar=cigar, **a)


@dataclass
class Pileup:
    status: int
    ref: np.ndarray
    pos: np.ndarray
    cov: np.ndarray
    nstart: np.ndarray
    refbase: np.ndarray
    counts: np.ndarray       # [n, 6]: A, C, G, T, other, '-'
    entry_off: np.ndarray | None
    entries: np.ndarray | None
    sum_coverage: int
    max_coverage: int
    pileup_ref_id: int

    @property
    def n_columns(self):
        return len(self.pos)

    def bases(self, k: int) -> str:
        e = self.entries[int(self.entry_off[k]):int(self.entry_off[k + 1])]
        return bytes(e["base"]).decode("latin-1")


def _take(res: _PileupResult, want_entries: bool) -> Pileup:
    n = int(res.n_columns)

    def arr(ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)

    p = Pileup(status=int(res.status), ref=arr(res.col_ref, n, np.int32), pos=arr(res.col_pos, n, np.uint64),
               cov=arr(res.col_cov, n, np.uint32), nstart=arr(res.col_nstart, n, np.uint32),
               refbase=arr(res.col_refbase, n, np.uint8), counts=arr(res.col_counts, n * 6, np.uint32).reshape(n, 6),
               entry_off=None, entries=None, sum_coverage=int(res.sum_coverage), max_coverage=int(res.max_coverage),
               pileup_ref_id=int(res.pileup_ref_id))
    if want_entries:
        p.entry_off = arr(res.col_entry_off, n + 1, np.uint64) if n else np.zeros(1, np.uint64)
        ne = int(res.n_entries)
        if ne:
            raw = C.string_at(res.entries, ne * C.sizeof(_Entry))
            p.entries = np.frombuffer(raw, dtype=ENTRY_DTYPE).copy()
        else:
            p.entries = np.zeros(0, ENTRY_DTYPE)
    lib().orc_pileup_free(C.byref(res))
    return p


def pileup(stream: bytes, mode=MODE_PILEUP_COLUMNS, use_md=False, start_from=0, end_at=U64_MAX, skip_zero=True,
           want_entries=False, allow=(ORC_OK, ORC_ESEQ)) -> Pileup:
    res = _PileupResult()
    rc = lib().orc_pileup(stream, len(stream), mode, int(use_md), start_from, end_at, int(skip_zero), int(want_entries), C.byref(res))
    if rc not in allow:
        lib().orc_pileup_free(C.byref(res))
        raise OracleError(rc)
    return _take(res, want_entries)


def pileup_chunked(stream: bytes, chunk_bytes=16384000, n_threads=2, want_entries=False, allow=(ORC_OK, ORC_ESEQ)) -> Pileup:
    res = _PileupResult()
    rc = lib().orc_pileup_chunked(stream, len(stream), chunk_bytes, n_threads, int(want_entries), C.byref(res))
    if rc not in allow:
        lib().orc_pileup_free(C.byref(res))
        raise OracleError(rc)
    return _take(res, want_entries)


class _Chunk(C.Structure):
    _fields_ = [("first", C.c_uint64), ("last", C.c_uint64), ("ref_id", C.c_int32), ("_pad", C.c_int32), ("beg", C.c_uint64), ("end", C.c_uint64)]


def chunk_plan(stream: bytes, chunk_bytes=16384000):
    """[(first, last, ref_id, beg, end)] of the chunks pileupChunks yields pileups for (start_from 0, end_at max)."""
    f = lib().orc_chunk_plan
    f.restype = C.c_int64; f.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(_Chunk), C.c_uint64]
    n = f(stream, len(stream), chunk_bytes, None, 0)
    if n < 0:
        raise OracleError(int(n))
    arr = (_Chunk * max(int(n), 1))()
    f(stream, len(stream), chunk_bytes, arr, n)
    return [(int(c.first), int(c.last), int(c.ref_id), int(c.beg), int(c.end)) for c in arr[:int(n)]]


def pileup_bench(stream, n_threads=1, chunk_bytes=16384000):
    r = _BenchResult()
    if isinstance(stream, np.ndarray):
        ptr = C.cast(stream.ctypes.data, C.c_char_p)
        n = stream.nbytes
    else:
        ptr, n = stream, len(stream)
    lib().orc_pileup_bench(ptr, n, n_threads, chunk_bytes, C.byref(r))
    return r


def md_operations(md: str):
    b = md.encode()
    n = lib().orc_md_operations(b, len(b), None, 0)
    ops = (_MdOp * max(n, 1))()
    lib().orc_md_operations(b, len(b), ops, n)
    out = []
    for o in ops[:n]:
        if o.type == 0:
            out.append(("match", int(o.match)))
        elif o.type == 1:
            out.append(("mismatch", o.mismatch.decode()))
        else:
            out.append(("deletion", md[o.del_off:o.del_off + o.del_len]))
    return out


def dna_read(stream: bytes, rec_off: int = 0) -> str:
    buf = C.create_string_buffer(1 << 20)
    n = lib().orc_dna_read(stream, rec_off, buf, len(buf))
    if n < 0:
        raise OracleError(int(n))
    return buf.raw[:n].decode()


def dna_reads(stream: bytes) -> str:
    buf = C.create_string_buffer(1 << 22)
    n = lib().orc_dna_reads(stream, len(stream), buf, len(buf))
    if n < 0:
        raise OracleError(int(n))
    return buf.raw[:n].decode()
