"""ctypes binding of libbiodgpu.so (the C ABI in include/biodgpu.h).

The library is the product; there is no Python or CPU implementation behind these calls.  If the
shared object is missing, importing :func:`lib` raises -- loudly -- instead of falling back.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("BIODGPU_SO") or os.path.join(_HERE, "libbiodgpu.so")      # (BIODGPU_SO: A/B builds of the same library for kernel experiments)

BGP_OK, BGP_EOF, BGP_EAGAIN = 0, 1, 2
BGP_EINVAL, BGP_ETRUNC, BGP_ECIGAR, BGP_EDEVICE, BGP_ENOMEM, BGP_ESTATE, BGP_EMD, BGP_EUNSORTED, BGP_ESEQ = -1, -2, -3, -4, -5, -6, -7, -8, -9
BGP_FETCH_SEQ, BGP_FETCH_QUAL, BGP_FETCH_CIGAR, BGP_FETCH_ALL = 1, 2, 4, 7
BGP_FETCH_STREAM = 8
BGP_DECODE_ONLY = 16        # bgp_flush_decode: no pileup will follow (the pass writes nothing for K4)
BGP_PILEUP_COLUMNS, BGP_MAKE_PILEUP = 0, 1
BGP_DEPTH_BINS = 65536
BGP_SNP_NO, BGP_SNP_YES, BGP_SNP_TIE, BGP_SNP_NEEDS_ENTRIES = 0, 1, 2, 3
BGP_MAQ_CALLED, BGP_MAQ_VARIANT, BGP_MAQ_SUBSAMPLED = 1, 2, 4
BGP_UOFFSET_UNKNOWN = 0xFFFFFFFF

RECORD_DTYPE = np.dtype([("ref_id", "<i4"), ("pos", "<i4"), ("end_pos", "<i4"), ("bin_mq_nl", "<u4"),
                         ("flag_nc", "<u4"), ("l_seq", "<i4"), ("rec_off", "<u8")])
COLUMN_DTYPE = np.dtype([("coverage", "<u4"), ("n_starting_here", "<u4"), ("a", "<u4"), ("c", "<u4"),
                         ("g", "<u4"), ("t", "<u4"), ("other", "<u4"), ("del", "<u4")])
MAQ_CALL_DTYPE = np.dtype([("quality", "<f4"), ("genotype", "u1"), ("reference_base5", "u1"), ("flags", "u1"), ("n_bases", "u1")])
ENTRY_DTYPE = np.dtype([("read_ordinal", "<u4"), ("cigar_op_offset", "<u4"), ("query_offset", "<u4"),
                        ("cigar_op_index", "<u2"), ("base", "u1"), ("qual", "u1")])
RUN_DTYPE = np.dtype([("ref_id", "<i4"), ("_pad", "<u4"), ("first_record", "<u8"), ("end_record", "<u8"),
                      ("start_pos", "<u8"), ("end_pos", "<u8"), ("column_base", "<u8")])
COLUMN8_DTYPE = np.dtype([(f, "u1") for f in COLUMN_DTYPE.names])      # bgp_column8: the same fields, a byte each (bgp_set_compact_columns)


def window_columns(cw, copy=True):
    """A window's columns as COLUMN_DTYPE whichever way they crossed the link: bgp_columns.columns, or columns8 widened (the consumer
    sees the same numbers, pileup.d:236-290)."""
    if cw.columns or not cw.columns8:
        return as_array(cw.columns, cw.n_positions, COLUMN_DTYPE, copy)
    c8 = as_array(cw.columns8, cw.n_positions, COLUMN8_DTYPE, False)
    out = np.empty(cw.n_positions, COLUMN_DTYPE)
    for f in COLUMN_DTYPE.names:
        out[f] = c8[f]
    return out


assert RECORD_DTYPE.itemsize == 32 and COLUMN_DTYPE.itemsize == 32 and ENTRY_DTYPE.itemsize == 16 and RUN_DTYPE.itemsize == 48


class Records(C.Structure):
    _fields_ = [("n", C.c_uint64), ("rec", C.c_void_p), ("seq_off", C.c_void_p), ("cigar_off", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("cigar", C.c_void_p), ("stream", C.c_void_p), ("stream_len", C.c_uint64)]


class PileupInfo(C.Structure):
    _fields_ = [("n_runs", C.c_uint32), ("runs", C.c_void_p), ("n_positions", C.c_uint64), ("n_columns", C.c_uint64),
                ("sum_coverage", C.c_uint64), ("max_coverage", C.c_uint32), ("soft_status", C.c_int32),
                ("n_reads_used", C.c_uint64), ("device_columns", C.c_void_p), ("device_depth_hist", C.c_void_p)]


class Columns(C.Structure):
    _fields_ = [("ref_id", C.c_int32), ("n_positions", C.c_uint32), ("first_position", C.c_uint64), ("columns", C.c_void_p),
                ("entry_off", C.c_void_p), ("entries", C.c_void_p), ("reference_base", C.c_void_p), ("snp", C.c_void_p), ("maq", C.c_void_p), ("columns8", C.c_void_p)]


class ChunkInfo(C.Structure):
    _fields_ = [("ref_id", C.c_int32), ("_pad", C.c_uint32), ("first_record", C.c_uint64), ("last_record", C.c_uint64),
                ("start_position", C.c_uint64), ("end_position", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("total_ms", C.c_double), ("launches", C.c_uint64), ("algorithmic_bytes", C.c_uint64)]


class SynthCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_reads", C.c_uint64), ("n_ref", C.c_uint32), ("ref_len", C.c_uint32),
                ("len_min", C.c_uint32), ("len_max", C.c_uint32), ("profile", C.c_uint32), ("first_ref", C.c_uint32)]


# every symbol include/biodgpu.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "bgp_strerror": (C.c_char_p, [C.c_int]),
    "bgp_abi_version": (C.c_int, []),
    "bgp_last_error": (C.c_char_p, [_P]),
    "bgp_init": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "bgp_destroy": (None, [_P]),
    "bgp_set_stream": (C.c_int, [_P, _P]),
    "bgp_set_n_ref": (C.c_int, [_P, C.c_int32]),
    "bgp_reset": (C.c_int, [_P]),
    "bgp_push_inflated": (C.c_int, [_P, _P, C.c_size_t, C.c_uint64, C.c_uint32]),
    "bgp_attach_device_stream": (C.c_int, [_P, _P, C.c_uint64]),
    "bgp_load_bam": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "bgp_load_bam_device": (C.c_int, [_P, C.c_char_p]),
    "bgp_n_ref": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bgp_ref_info": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]),
    "bgp_flush": (C.c_int, [_P]),
    "bgp_flush_decode": (C.c_int, [_P, C.c_uint32]),
    "bgp_num_records": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "bgp_decode_fetch": (C.c_int, [_P, C.c_uint32, C.POINTER(Records)]),
    "bgp_decode_device": (C.c_int, [_P, C.c_uint32, C.POINTER(Records)]),
    "bgp_decode_fetch_range": (C.c_int, [_P, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Records)]),
    "bgp_pileup_run": (C.c_int, [_P, C.POINTER(PileupInfo)]),
    "bgp_pileup_begin": (C.c_int, [_P, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint32]),
    "bgp_pileup_begin_ref": (C.c_int, [_P, C.c_int32, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint32]),
    "bgp_pileup_next_window": (C.c_int, [_P, C.POINTER(Columns)]),
    "bgp_pileup_ref_id": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bgp_pileup_snp": (C.c_int, [_P, C.c_int, C.c_int]),
    "bgp_pileup_maq": (C.c_int, [_P, C.c_float, C.c_float, C.c_uint32]),
    "bgp_maq_error_model": (C.c_int, [C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgp_batch_begin": (C.c_int, [_P, C.c_uint64, C.c_int, C.c_uint32]),
    "bgp_stream_bam_begin": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_int, C.c_uint32]),
    "bgp_device_memory": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    "bgp_set_compact_columns": (C.c_int, [_P, C.c_int]),
    "bgp_set_read_filter": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "bgp_set_device_memory_cap": (C.c_int, [C.c_uint64]),
    "bgp_batch_push": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "bgp_batch_push_pinned": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "bgp_batch_set_md": (C.c_int, [_P, C.c_int, C.c_int]),
    "bgp_batch_acquire": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "bgp_batch_commit": (C.c_int, [_P, C.c_size_t]),
    "bgp_batch_end": (C.c_int, [_P]),
    "bgp_batch_next_window": (C.c_int, [_P, C.POINTER(Columns)]),
    "bgp_batch_ordinal_base": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "bgp_pileup_chunks_begin": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]),
    "bgp_pileup_chunk": (C.c_int, [_P, C.c_uint64, C.POINTER(ChunkInfo)]),
    "bgp_read_dna": (C.c_int, [_P, C.c_uint64, C.c_char_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "bgp_bai_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "bgp_bai_close": (None, [_P]),
    "bgp_bai_n_ref": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "bgp_bai_shape": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bgp_bai_min_offset": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "bgp_bai_chunks": (C.c_int, [_P, C.c_int32, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint32)]),
    "bgp_reg2bin": (C.c_int, [C.c_int32, C.c_int32]),
    "bgp_bin_can_overlap": (C.c_int, [C.c_uint32, C.c_int32, C.c_int32]),
    "bgp_bgzf_read_chunks": (C.c_int, [C.c_char_p, _P, C.c_uint32, C.c_int, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bgp_load_bam_region": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_uint32, C.c_uint32, C.c_int]),
    "bgp_load_bam_region_device": (C.c_int, [_P, C.c_char_p, _P, C.c_int32, C.c_uint32, C.c_uint32]),
    "bgp_filter_region": (C.c_int, [_P, C.c_int32, C.c_uint32, C.c_uint32]),
    "bgp_load_bam_read_at": (C.c_int, [_P, C.c_char_p, C.c_uint64]),
    "bgp_record_voffset": (C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bgp_bai_build": (C.c_int, [_P, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint64)]),
    "bgp_hist_reduce": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "bgp_timing_enable": (C.c_int, [_P, C.c_int]),
    "bgp_timing_reset": (C.c_int, [_P]),
    "bgp_timing_get": (C.c_int, [_P, C.POINTER(C.POINTER(KernelTime)), C.POINTER(C.c_uint32)]),
    "bgp_synth_host": (C.c_int, [C.POINTER(SynthCfg), C.c_uint64, C.c_uint64, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "bgp_synth_device": (C.c_int, [_P, C.POINTER(SynthCfg), C.POINTER(C.c_uint64)]),
}

_lib = None


class BiodGpuError(Exception):
    """Raised for every negative status of the C ABI (the D facade rethrows these as Exception)."""

    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = lib().bgp_strerror(code).decode()
        super().__init__(f"libbiodgpu status {code}: {msg}" + (f" [{detail}]" if detail else ""))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback for the decode/pileup kernels)")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError here = the .so does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc: int, ctx=None) -> int:
    if rc < 0:
        detail = ""
        try:
            d = lib().bgp_last_error(ctx)
            detail = d.decode() if d else ""
        except Exception:
            pass
        raise BiodGpuError(rc, detail)
    return rc


def as_array(ptr, count, dtype, copy=True):
    """`count` items of `dtype` from a host pointer owned by the ctx: a copy, or (copy=False) a view of the ctx's own
    pinned buffer -- its CONTENT is valid until the next call that hands out a window, as the C ABI says; the memory
    itself stays mapped until bgp_reset / bgp_destroy (a buffer a larger window replaces is retired, not freed), so a
    view kept too long reads stale bytes, never freed memory."""
    count = int(count)
    if not ptr or count == 0:
        return np.zeros(0, dtype)
    dtype = np.dtype(dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    a = np.frombuffer(buf, dtype=dtype, count=count)
    return a.copy() if copy else a
