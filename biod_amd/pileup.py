"""Host-side mirror of BioD's pileup interface over the libbiodgpu C ABI.

Names, argument order, defaults and error behaviour follow ``bio/std/hts/bam/pileup.d``:
``makePileup(reads, use_md_tag=false, start_from=0, end_at=ulong.max, skip_zero_coverage=true)``
(pileup.d:683-687) and ``pileupColumns(reads, use_md_tag=false, skip_zero_coverage=true)``
(pileup.d:509); columns expose ``coverage / ref_id / position / reads / reads_starting_here / bases /
base_qualities / read_qualities`` (pileup.d:236-290) and every read of a column exposes
``cigar_operation / cigar_operation_offset / cigar_before / cigar_after / current_base /
current_base_quality / query_offset`` (pileup.d:86-230) plus the ``BamRead`` accessors used by BioD's
consumers (name, position, mapping_quality, is_reverse_strand, sequence, base_qualities, cigar).

``reads`` is a :class:`RecordStream` (inflated record bytes, host or device resident) or a
:class:`BamReader` (file), the analogue of ``BamReader.reads`` (reader.d:228-231).  All computation
happens in the HIP kernels behind the ABI; this module only moves pointers and wraps results.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional

import numpy as np

from . import _lib as L

ULONG_MAX = (1 << 64) - 1
_CIGAR_CHARS = "MIDNSHP=X???????"


class Context:
    """One bgp_ctx: a (host thread, GPU) pair."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        L.check(L.lib().bgp_init(device, C.byref(self._h)))
        self._keep = None

    def close(self):
        if self._h:
            L.lib().bgp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- staging
    def reset(self):
        L.check(L.lib().bgp_reset(self._h), self._h)
        self._keep = None

    def set_stream(self, hip_stream: int):
        L.check(L.lib().bgp_set_stream(self._h, C.c_void_p(hip_stream)), self._h)

    def set_n_ref(self, n_ref: int):
        L.check(L.lib().bgp_set_n_ref(self._h, n_ref), self._h)

    def push_inflated(self, data: bytes, coffset: int = 0, first_record_uoffset: int = L.BGP_UOFFSET_UNKNOWN):
        buf = (C.c_char * len(data)).from_buffer_copy(data) if data else None
        L.check(L.lib().bgp_push_inflated(self._h, C.cast(buf, C.c_void_p) if buf else None, len(data), coffset, first_record_uoffset), self._h)

    def attach_device_stream(self, device_ptr: int, length: int, keepalive=None):
        self._keep = keepalive
        L.check(L.lib().bgp_attach_device_stream(self._h, C.c_void_p(device_ptr), length), self._h)

    def load_bam(self, path: str, n_threads: int = 0):
        L.check(L.lib().bgp_load_bam(self._h, path.encode(), n_threads), self._h)

    def load_bam_device(self, path: str):
        """bgp_load_bam with decompressBgzfBlock (block.d:127-216) on the device; no host copy of the stream."""
        L.check(L.lib().bgp_load_bam_device(self._h, path.encode()), self._h)

    def load_bam_region(self, path: str, bai: "BaiFile", ref_id: int, beg: int, end: int, n_threads: int = 0):
        """bam[ref][beg .. end] (reference.d:73-81): stage the BAI chunks' records; flush() then keeps the overlapping reads."""
        L.check(L.lib().bgp_load_bam_region(self._h, path.encode(), bai.handle, ref_id, beg, end, n_threads), self._h)

    def load_bam_region_device(self, path: str, bai: "BaiFile", ref_id: int, beg: int, end: int):
        """load_bam_region with the blocks inflated on the device (bgp_load_bam_region_device)."""
        L.check(L.lib().bgp_load_bam_region_device(self._h, path.encode(), bai.handle, ref_id, beg, end), self._h)

    def filter_region(self, ref_id: int, beg: int, end: int):
        """BamReadFilter (randomaccessmanager.d:366-462) on the flushed record table."""
        L.check(L.lib().bgp_filter_region(self._h, ref_id, beg, end), self._h)

    def load_bam_read_at(self, path: str, voffset: int):
        """getReadAt (randomaccessmanager.d:156-164)."""
        L.check(L.lib().bgp_load_bam_read_at(self._h, path.encode(), voffset), self._h)

    def record_voffset(self, record: int) -> int:
        """read.start_virtual_offset (reads!withOffsets, readrange.d:64-86)."""
        v = C.c_uint64()
        L.check(L.lib().bgp_record_voffset(self._h, record, C.byref(v)), self._h)
        return v.value

    def bai_build(self) -> bytes:
        """createIndex (bai/indexing.d:353-366): the .bai bytes of the one whole file this context staged."""
        p, n = C.POINTER(C.c_uint8)(), C.c_uint64()
        L.check(L.lib().bgp_bai_build(self._h, C.byref(p), C.byref(n)), self._h)
        return C.string_at(p, n.value)

    def staged_bytes(self) -> np.ndarray:
        """Host copy of the staged (inflated) record bytes; record i starts at rec['rec_off'][i]."""
        r = L.Records()
        L.check(L.lib().bgp_decode_fetch(self._h, L.BGP_FETCH_STREAM, C.byref(r)), self._h)
        return L.as_array(r.stream, r.stream_len, "u1")

    def set_read_filter(self, flag_require: int = 0, flag_exclude: int = 0, min_mapping_quality: int = 0):
        """reads.filter!(...) in front of the pileup (bgp_set_read_filter): reads that fail take part in nothing from the next flush on."""
        L.check(L.lib().bgp_set_read_filter(self._h, flag_require, flag_exclude, min_mapping_quality), self._h)

    def set_compact_columns(self, on: bool = True):
        """Windows cross the link as 8 bytes a position wherever every count fits a byte (bgp_set_compact_columns); this facade widens them."""
        L.check(L.lib().bgp_set_compact_columns(self._h, int(on)), self._h)

    def references(self):
        n = C.c_int32()
        L.check(L.lib().bgp_n_ref(self._h, C.byref(n)), self._h)
        out = []
        for i in range(n.value):
            nm, ln = C.c_char_p(), C.c_int32()
            L.check(L.lib().bgp_ref_info(self._h, i, C.byref(nm), C.byref(ln)), self._h)
            out.append((nm.value.decode("latin-1"), ln.value))
        return out

    def synth_device(self, cfg: "L.SynthCfg") -> int:
        n = C.c_uint64()
        L.check(L.lib().bgp_synth_device(self._h, C.byref(cfg), C.byref(n)), self._h)
        return n.value

    # -- decode
    def flush(self, decode: int = 0):
        """Framing + record table; with `decode` (BGP_FETCH_* bits) the arenas are made in the same pass over the stream."""
        if decode:
            L.check(L.lib().bgp_flush_decode(self._h, decode), self._h)
        else:
            L.check(L.lib().bgp_flush(self._h), self._h)

    def num_records(self) -> int:
        n = C.c_uint64()
        L.check(L.lib().bgp_num_records(self._h, C.byref(n)), self._h)
        return n.value

    def decode_fetch(self, what: int = L.BGP_FETCH_ALL) -> "DecodedRecords":
        r = L.Records()
        L.check(L.lib().bgp_decode_fetch(self._h, what, C.byref(r)), self._h)
        n = r.n
        seq_off = L.as_array(r.seq_off, n + 1, "<u8")
        cigar_off = L.as_array(r.cigar_off, n + 1, "<u8")
        return DecodedRecords(rec=L.as_array(r.rec, n, L.RECORD_DTYPE), seq_off=seq_off, cigar_off=cigar_off,
                              seq=L.as_array(r.seq, int(seq_off[n]), "u1") if r.seq else None,
                              qual=L.as_array(r.qual, int(seq_off[n]), "u1") if r.qual else None,
                              cigar=L.as_array(r.cigar, int(cigar_off[n]), "<u4") if r.cigar else None)

    def decode_fetch_range(self, first: int, count: int, what: int = L.BGP_FETCH_ALL, stream: bool = False) -> "DecodedRecords":
        """Records [first, first + count): rows, arena slices (offsets keep their stream-wide values) and, with
        stream=True, the records' own bytes (names, tags) as `.data`, record i at rec_off[i] - rec_off[0]."""
        r = L.Records()
        L.check(L.lib().bgp_decode_fetch_range(self._h, what | (L.BGP_FETCH_STREAM if stream else 0), first, count, C.byref(r)), self._h)
        n = r.n
        seq_off = L.as_array(r.seq_off, n + 1, "<u8")
        cigar_off = L.as_array(r.cigar_off, n + 1, "<u8")
        nb, nc = (int(seq_off[n] - seq_off[0]), int(cigar_off[n] - cigar_off[0])) if n else (0, 0)
        d = DecodedRecords(rec=L.as_array(r.rec, n, L.RECORD_DTYPE), seq_off=seq_off, cigar_off=cigar_off,
                           seq=L.as_array(r.seq, nb, "u1") if r.seq else None, qual=L.as_array(r.qual, nb, "u1") if r.qual else None,
                           cigar=L.as_array(r.cigar, nc, "<u4") if r.cigar else None)
        d.first = first
        d.data = L.as_array(r.stream, r.stream_len, "u1") if stream and r.stream else None
        return d

    def decode_device(self, what: int = L.BGP_FETCH_ALL) -> "L.Records":
        r = L.Records()
        L.check(L.lib().bgp_decode_device(self._h, what, C.byref(r)), self._h)
        return r

    # -- pileup
    def pileup_run(self) -> "PileupInfo":
        info = L.PileupInfo()
        L.check(L.lib().bgp_pileup_run(self._h, C.byref(info)), self._h)
        return PileupInfo(runs=L.as_array(info.runs, info.n_runs, L.RUN_DTYPE), n_positions=info.n_positions, n_columns=info.n_columns,
                          sum_coverage=info.sum_coverage, max_coverage=info.max_coverage, soft_status=info.soft_status,
                          n_reads_used=info.n_reads_used, device_columns=info.device_columns or 0, device_depth_hist=info.device_depth_hist or 0)

    def pileup_windows(self, mode: int, use_md_tag=False, start_from=0, end_at=ULONG_MAX, skip_zero_coverage=True,
                       want_entries=False, window_positions=0, snp=None, maq=None, copy=True, ref_id=None):
        """Dense windows of columns (copy=False: the arrays are views of the ctx's pinned buffers, valid until the next
        window -- BioD's own warning about column.reads, pileup.d:660-661 -- and a window costs no host copy).  snp=(minimum_coverage, minimum_witnesses) adds simple.isSNP flags per position
        (snpcallers/simple.d:15-41, needs use_md_tag): window.snp, resolved against the entries where the device says
        BGP_SNP_NEEDS_ENTRIES and entries were requested.  maq=(depcorr, eta, minimum_base_quality) adds the MAQ model's
        call per position (snpcallers/maq.d:388-486): window.maq, a MAQ_CALL_DTYPE array."""
        if ref_id is not None:       # makePileup over the reads of one reference of the stream (bgp_pileup_begin_ref)
            L.check(L.lib().bgp_pileup_begin_ref(self._h, int(ref_id), int(use_md_tag), start_from, end_at, int(skip_zero_coverage), int(want_entries), window_positions), self._h)
        else:
            L.check(L.lib().bgp_pileup_begin(self._h, mode, int(use_md_tag), start_from, end_at, int(skip_zero_coverage), int(want_entries), window_positions), self._h)
        if snp is not None:
            L.check(L.lib().bgp_pileup_snp(self._h, int(snp[0]), int(snp[1])), self._h)
        if maq is not None:
            L.check(L.lib().bgp_pileup_maq(self._h, float(maq[0]), float(maq[1]), int(maq[2])), self._h)
        yield from self._windows(snp, copy)

    def pileup_chunks_begin(self, use_md_tag=False, block_size=16_384_000, start_from=0, end_at=ULONG_MAX, want_entries=False, window_positions=0) -> int:
        """Plans pileupChunks (pileup.d:1011-1015); returns the number of chunks."""
        n = C.c_uint64()
        L.check(L.lib().bgp_pileup_chunks_begin(self._h, int(use_md_tag), block_size, start_from, end_at, int(want_entries), window_positions, C.byref(n)), self._h)
        return int(n.value)

    def pileup_chunk(self, k: int) -> "L.ChunkInfo":
        """Selects chunk k: its windows follow from chunk_windows()."""
        ci = L.ChunkInfo()
        L.check(L.lib().bgp_pileup_chunk(self._h, k, C.byref(ci)), self._h)
        return ci

    def chunk_windows(self, k: int, copy=True):
        self.pileup_chunk(k)
        yield from self._windows(None, copy)

    # -- batched staging (streams larger than HBM): include/biodgpu.h, bgp_batch_*
    def batch_begin(self, batch_bytes: int = 0, want_entries: bool = False, window_positions: int = 0, maq=None, use_md_tag=False, skip_zero_coverage=True):
        """maq=(depcorr, eta, minimum_base_quality): the MAQ model's call per position on every batch's windows (window.maq).
        use_md_tag: window.reference_base, the provider chain carried across the batches (bgp_batch_set_md)."""
        L.check(L.lib().bgp_batch_begin(self._h, batch_bytes, int(want_entries), window_positions), self._h)
        if use_md_tag:
            L.check(L.lib().bgp_batch_set_md(self._h, 1, int(skip_zero_coverage)), self._h)
        if maq is not None:
            L.check(L.lib().bgp_pileup_maq(self._h, float(maq[0]), float(maq[1]), int(maq[2])), self._h)

    def batch_push(self, data) -> tuple:
        """Returns (bytes taken, status): status BGP_EAGAIN = hand out windows first, then push the rest."""
        mv = memoryview(data)
        buf = (C.c_char * len(mv)).from_buffer_copy(mv) if len(mv) else None
        taken = C.c_size_t()
        rc = L.check(L.lib().bgp_batch_push(self._h, buf, len(mv), C.byref(taken)), self._h)
        return int(taken.value), rc

    def batch_acquire(self):
        """(writable view of the pinned buffer's free tail, status): the zero-copy side of batch_push (bgp_batch_acquire).  BGP_EAGAIN
        (view None): hand out windows first.  The view is the caller's until batch_commit(n)."""
        ptr = C.c_void_p(); cap = C.c_size_t()
        rc = L.check(L.lib().bgp_batch_acquire(self._h, C.byref(ptr), C.byref(cap)), self._h)
        if rc != L.BGP_OK or not ptr.value:
            return None, rc
        return memoryview((C.c_char * cap.value).from_address(ptr.value)).cast("B"), rc

    def batch_commit(self, n: int):
        L.check(L.lib().bgp_batch_commit(self._h, n), self._h)

    def batch_end(self):
        L.check(L.lib().bgp_batch_end(self._h), self._h)

    def batch_ordinal_base(self) -> int:
        n = C.c_uint64()
        L.check(L.lib().bgp_batch_ordinal_base(self._h, C.byref(n)), self._h)
        return int(n.value)

    def batch_drain(self, limit=None):
        """Windows of final columns until the library wants more input (or, after batch_end, until the stream is done); at most
        `limit` of them.  (Bytes and ONE window in turn keep a full pinned buffer queued whenever a device buffer frees, and both
        directions of the link busy: INTEGRATION.md.)"""
        while limit is None or limit > 0:
            if limit is not None:
                limit -= 1
            cw = L.Columns()
            rc = L.check(L.lib().bgp_batch_next_window(self._h, C.byref(cw)), self._h)
            if rc != L.BGP_OK:
                return
            cols = L.window_columns(cw)
            eoff = L.as_array(cw.entry_off, cw.n_positions + 1, "<u8") if cw.entry_off else None
            ents = L.as_array(cw.entries, int(eoff[-1]), L.ENTRY_DTYPE) if eoff is not None else None
            w = Window(cw.ref_id, cw.first_position, cols, eoff, ents, L.as_array(cw.reference_base, cw.n_positions, "u1") if cw.reference_base else None)
            if cw.maq:
                w.maq = L.as_array(cw.maq, cw.n_positions, L.MAQ_CALL_DTYPE)
            yield w

    def stream_bam_windows(self, path: str, batch_bytes: int = 0, want_entries: bool = False, window_positions: int = 0, maq=None, use_md_tag=False, skip_zero_coverage=True):
        """The windows of pileupColumns(new BamReader(path).reads) with the FILE streamed through the device in bounded memory
        (bgp_stream_bam_begin: reader threads -> compressed ring in HBM -> k_inflate -> a batch's kernels -> windows)."""
        L.check(L.lib().bgp_stream_bam_begin(self._h, path.encode(), batch_bytes, int(want_entries), window_positions), self._h)
        if use_md_tag:
            L.check(L.lib().bgp_batch_set_md(self._h, 1, int(skip_zero_coverage)), self._h)
        if maq is not None:
            L.check(L.lib().bgp_pileup_maq(self._h, float(maq[0]), float(maq[1]), int(maq[2])), self._h)
        yield from self.batch_drain()

    def batched_windows(self, pieces, batch_bytes: int = 0, want_entries: bool = False, window_positions: int = 0, maq=None, use_md_tag=False, skip_zero_coverage=True):
        """pileupColumns over a stream handed over piece by piece (any cut points): dense windows of final columns."""
        self.batch_begin(batch_bytes, want_entries, window_positions, maq, use_md_tag, skip_zero_coverage)
        for piece in pieces:
            mv = memoryview(piece)
            off = 0
            while off < len(mv):
                taken, rc = self.batch_push(mv[off:])
                off += taken
                yield from self.batch_drain(1)
        self.batch_end()
        yield from self.batch_drain()

    def batched_windows_zero_copy(self, pieces, batch_bytes: int = 0, want_entries: bool = False, window_positions: int = 0, use_md_tag=False, skip_zero_coverage=True):
        """batched_windows with the pieces written straight into the pinned staging buffer (bgp_batch_acquire / _commit)."""
        self.batch_begin(batch_bytes, want_entries, window_positions, None, use_md_tag, skip_zero_coverage)
        for piece in pieces:
            mv = memoryview(piece).cast("B")
            off = 0
            while off < len(mv):
                view, rc = self.batch_acquire()
                if rc == L.BGP_EAGAIN:
                    yield from self.batch_drain(1)
                    continue
                m = min(len(view), len(mv) - off)
                view[:m] = mv[off:off + m]
                self.batch_commit(m)
                off += m
                yield from self.batch_drain(1)
        self.batch_end()
        yield from self.batch_drain()

    def _windows(self, snp, copy=True):
        while True:
            cw = L.Columns()
            rc = L.check(L.lib().bgp_pileup_next_window(self._h, C.byref(cw)), self._h)
            if rc == L.BGP_EOF:
                return
            cols = L.window_columns(cw, copy)
            eoff = L.as_array(cw.entry_off, cw.n_positions + 1, "<u8", copy) if cw.entry_off else None
            ents = L.as_array(cw.entries, int(eoff[-1]), L.ENTRY_DTYPE, copy) if eoff is not None else None
            refb = L.as_array(cw.reference_base, cw.n_positions, "u1", copy) if cw.reference_base else None
            w = Window(cw.ref_id, cw.first_position, cols, eoff, ents, refb)
            if cw.maq:
                w.maq = L.as_array(cw.maq, cw.n_positions, L.MAQ_CALL_DTYPE, copy)
            if snp is not None and cw.snp:
                w.snp = L.as_array(cw.snp, cw.n_positions, "u1")
                if ents is not None:
                    for k in np.nonzero(w.snp == L.BGP_SNP_NEEDS_ENTRIES)[0]:
                        w.snp[k] = is_snp_from_bases(bytes(ents["base"][int(eoff[k]):int(eoff[k + 1])]), int(cols["coverage"][k]), int(refb[k]), int(snp[0]), int(snp[1]))
            yield w

    def read_dna(self, record: int, cap: int = 1 << 16, strict: bool = True) -> str:
        """dna(read), md/reconstruct.d:38-214: reference bases spanned by a read, from CIGAR + MD.
        With strict=False the chars BioD's lazy range yields before it stalls are returned instead of
        raising BGP_EMD."""
        buf = C.create_string_buffer(max(cap, 1))
        n = C.c_uint32()
        rc = L.lib().bgp_read_dna(self._h, record, buf, cap, C.byref(n))
        if rc != L.BGP_EMD or strict:
            L.check(rc, self._h)
        if n.value > cap:
            return self.read_dna(record, n.value, strict)
        return buf.raw[:n.value].decode("latin-1")

    def pileup_ref_id(self) -> int:
        r = C.c_int32()
        L.check(L.lib().bgp_pileup_ref_id(self._h, C.byref(r)), self._h)
        return r.value

    # -- timing
    def timing_enable(self, on=True):
        L.check(L.lib().bgp_timing_enable(self._h, int(on)), self._h)

    def timing_reset(self):
        L.check(L.lib().bgp_timing_reset(self._h), self._h)

    def timing_get(self):
        p = C.POINTER(L.KernelTime)()
        n = C.c_uint32()
        L.check(L.lib().bgp_timing_get(self._h, C.byref(p), C.byref(n)), self._h)
        return {p[i].name.decode(): dict(total_ms=p[i].total_ms, launches=p[i].launches, algorithmic_bytes=p[i].algorithmic_bytes) for i in range(n.value)}

    @property
    def handle(self):
        return self._h


class DecodedRecords:
    def __init__(self, rec, seq_off, cigar_off, seq, qual, cigar):
        self.rec, self.seq_off, self.cigar_off, self.seq, self.qual, self.cigar = rec, seq_off, cigar_off, seq, qual, cigar

    def __len__(self):
        return len(self.rec)


class PileupInfo:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def is_snp_from_bases(bases: bytes, coverage: int, reference_base: int, minimum_coverage: int = 5, minimum_witnesses: int = 2) -> int:
    """simple.isSNP (snpcallers/simple.d:15-41) on the current_base chars of one column; BGP_SNP_TIE where BioD's
    answer hangs on the iteration order of druntime's associative array (several bases share the largest count and
    reference_base is one of them)."""
    if coverage < minimum_coverage % (1 << 64):         # size_t < int: D converts the int (simple.d:17)
        return L.BGP_SNP_NO
    cnt = {}
    for b in bases:
        if b != 0x2D:                                   # '-'
            cnt[b] = cnt.get(b, 0) + 1
    if not cnt:
        return L.BGP_SNP_NO
    top = max(cnt.values())
    if top < minimum_witnesses:
        return L.BGP_SNP_NO
    tops = [b for b, v in cnt.items() if v == top]
    if reference_base not in tops:
        return L.BGP_SNP_YES
    return L.BGP_SNP_NO if len(tops) == 1 else L.BGP_SNP_TIE


class Window:
    def __init__(self, ref_id, first_position, columns, entry_off, entries, reference_base=None):
        self.ref_id, self.first_position, self.columns, self.entry_off, self.entries = ref_id, first_position, columns, entry_off, entries
        self.reference_base = reference_base          # uint8 per position (use_md_tag) or None
        self.snp = None                               # uint8 per position of BGP_SNP_* (pileup_windows(snp=...)) or None
        self.maq = None                               # MAQ_CALL_DTYPE per position (pileup_windows(maq=...)) or None


# ------------------------------------------------------------------------------------------ reads

class RecordStream:
    """An input range of BAM reads given as inflated record bytes (what ``BamReadRange`` consumes,
    readrange.d:118-173).  Holds the host copy so per-read accessors can view names and tags."""

    def __init__(self, data: bytes, ctx: Optional[Context] = None, n_ref: int = 0):
        self.data = data
        self.ctx = ctx or Context()
        self.n_ref = n_ref
        self._staged = False
        self.records = None

    def _stage(self, decode: int = 0):
        if not self._staged:
            self.ctx.reset()
            if self.n_ref:
                self.ctx.set_n_ref(self.n_ref)
            self.ctx.push_inflated(self.data)
            self.ctx.flush(decode)        # (decode != 0: the arenas in the same pass over the stream, bgp_flush_decode)
            self._staged = True
        return self.ctx

    def decode(self, what=L.BGP_FETCH_ALL) -> DecodedRecords:
        self.records = self._stage(what).decode_fetch(what)
        return self.records


class BaiFile:
    """``BaiFile(filename)`` (baifile.d:87-169) -- host side, no GPU needed."""

    def __init__(self, filename: str):
        self._h = C.c_void_p()
        L.check(L.lib().bgp_bai_open(filename.encode(), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            L.lib().bgp_bai_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_ref(self) -> int:
        n = C.c_int32()
        L.check(L.lib().bgp_bai_n_ref(self._h, C.byref(n)))
        return n.value

    def shape(self, ref_id: int):
        """(bins, chunks, linear-index intervals) of one reference."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.check(L.lib().bgp_bai_shape(self._h, ref_id, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def min_offset(self, ref_id: int, position: int) -> int:
        """Index.getMinimumOffset, baifile.d:77-82."""
        v = C.c_uint64()
        L.check(L.lib().bgp_bai_min_offset(self._h, ref_id, position, C.byref(v)))
        return v.value

    def chunks(self, ref_id: int, beg: int, end: int) -> np.ndarray:
        """RandomAccessManager.getChunks, randomaccessmanager.d:222-244: (n, 2) uint64 virtual offsets."""
        p, n = C.c_void_p(), C.c_uint32()
        L.check(L.lib().bgp_bai_chunks(self._h, ref_id, beg, end, C.byref(p), C.byref(n)))
        return L.as_array(p.value, 2 * n.value, "<u8").reshape(-1, 2)


def read_chunks(bam_path: str, chunks: np.ndarray, n_threads: int = 0) -> bytes:
    """The inflated bytes a chunk list stands for (StreamChunksSupplier, inputstream.d:254-330)."""
    ch = np.ascontiguousarray(chunks, dtype="<u8").reshape(-1, 2)
    n = C.c_uint64()
    L.check(L.lib().bgp_bgzf_read_chunks(bam_path.encode(), ch.ctypes.data, len(ch), n_threads, None, 0, C.byref(n)))
    buf = np.zeros(max(n.value, 1), np.uint8)
    L.check(L.lib().bgp_bgzf_read_chunks(bam_path.encode(), ch.ctypes.data, len(ch), n_threads, buf.ctypes.data, n.value, C.byref(n)))
    return buf[:n.value].tobytes()


class _RegionReads:
    """``bam["chr1"][beg .. end]`` (reference.d:73-81): reads overlapping the region, staged through the BAI."""

    def __init__(self, reader: "BamReader", ref_id: int, beg: int, end: int):
        self.reader, self.ref_id, self.beg, self.end = reader, ref_id, beg, end
        self.data = None
        self.records = None
        self.ctx = None

    def _stage(self):
        r = self.reader
        ctx = r._region_context()
        if r._region_owner is not self:
            ctx.reset()
            ctx.load_bam_region(r.filename, r.bai, self.ref_id, self.beg, self.end, r.n_threads)
            ctx.flush()
            r._region_owner = self
        self.ctx = ctx
        return ctx

    def decode(self, what=L.BGP_FETCH_ALL):
        self.records = self._stage().decode_fetch(what)
        return self.records

    def __len__(self):
        return self._stage().num_records()

    @property
    def empty(self) -> bool:
        return len(self) == 0

    def names(self) -> List[str]:
        ctx = self._stage()
        rec = ctx.decode_fetch(0).rec
        data = ctx.staged_bytes()
        out = []
        for r in rec:
            off, ln = int(r["rec_off"]) + 36, int(r["bin_mq_nl"]) & 0xFF
            out.append(data[off:off + ln - 1].tobytes().decode("latin-1"))
        return out


class ReferenceSequence:
    """reference.d:40-221: ``bam["chr1"]``; slicing gives the reads overlapping ``[start, end)``."""

    def __init__(self, reader: "BamReader", ref_id: int, name: str, length: int):
        self._reader, self.id, self.name, self.length = reader, ref_id, name, length

    def __getitem__(self, sl) -> _RegionReads:
        if not isinstance(sl, slice) or sl.step is not None:
            raise TypeError("use bam[name][start:end]")
        start = 0 if sl.start is None else int(sl.start)
        end = self.length if sl.stop is None else int(sl.stop)
        return _RegionReads(self._reader, self.id, start, end)

    def reads(self) -> _RegionReads:                     # opSlice(), reference.d:84-86
        return self[0:self.length]

    def startVirtualOffset(self) -> int:                 # reference.d:96-104 (empty -> EOF offset is the host's business)
        rr = self.reads()
        if rr.empty:
            raise ValueError("no reads on this reference")
        return rr._stage().record_voffset(0)

    def firstPosition(self) -> int:                      # reference.d:147-153
        rr = self.reads()
        return -1 if rr.empty else int(rr._stage().decode_fetch(0).rec["pos"][0])


class BamReader:
    """``new BamReader(filename)`` (reader.d:127-138): ``.reads`` is the range handed to makePileup;
    ``reader["chr1"][a:b]`` (reader.d opIndex -> reference.d:73-81) needs the ``.bai`` next to the file."""

    def __init__(self, filename: str, n_threads: int = 0, ctx: Optional[Context] = None, device_inflate: bool = False):
        self.filename = filename
        self.n_threads = n_threads
        self.ctx = ctx or Context()
        self.ctx.reset()
        if device_inflate:                              # decompressBgzfBlock on the GPU (inflate.hip), for hosts with few cores
            self.ctx.load_bam_device(filename)
        else:
            self.ctx.load_bam(filename, n_threads)
        self.reference_sequences = self.ctx.references()
        self.ctx.flush()
        self.reads = _LoadedReads(self.ctx)
        self._bai = None
        self._rctx = None
        self._region_owner = None

    @property
    def bai(self) -> BaiFile:
        if self._bai is None:
            self._bai = BaiFile(self.filename)          # "BAM index file (.bai) must be provided", randomaccessmanager.d:201-203
        return self._bai

    @property
    def has_index(self) -> bool:                        # reader.d:150-152
        import os
        stem = self.filename.rsplit(".", 1)[0]             # baifile.d:92-121: x.bam.bai, then x.bai
        return os.path.exists(self.filename + ".bai") or os.path.exists(stem + ".bai")

    def createIndex(self, overwrite: bool = False, path: Optional[str] = None):
        """reader.d:158-168: writes filename + ".bai" unless an index is there already."""
        if self.has_index and not overwrite and path is None:
            return
        data = self.ctx.bai_build()
        with open(path or self.filename + ".bai", "wb") as f:
            f.write(data)
        self._bai = None

    def _region_context(self) -> Context:
        if self._rctx is None:
            self._rctx = Context()
        return self._rctx

    def __getitem__(self, name: str) -> ReferenceSequence:
        for i, (nm, ln) in enumerate(self.reference_sequences):
            if nm == name:
                return ReferenceSequence(self, i, nm, ln)
        raise KeyError(name)

    def reference(self, ref_id: int) -> ReferenceSequence:
        nm, ln = self.reference_sequences[ref_id]
        return ReferenceSequence(self, ref_id, nm, ln)

    def getReadAt(self, voffset: int) -> "RecordStream":
        """randomaccessmanager.d:156-164: the single read starting at a virtual offset."""
        ctx = self._region_context()
        ctx.reset()
        ctx.load_bam_read_at(self.filename, voffset)
        ctx.flush()
        self._region_owner = None
        rs = _LoadedReads(ctx)
        return rs


class _LoadedReads:
    def __init__(self, ctx):
        self.ctx = ctx
        self.data = None
        self.records = None

    def _stage(self):
        return self.ctx

    def decode(self, what=L.BGP_FETCH_ALL):
        self.records = self.ctx.decode_fetch(what)
        return self.records


# ------------------------------------------------------------------------------------------ columns

class CigarOperation:
    """cigar.d:58-136"""
    __slots__ = ("raw",)

    def __init__(self, raw: int):
        self.raw = int(raw)

    @property
    def length(self):
        return self.raw >> 4

    @property
    def type(self):
        return _CIGAR_CHARS[self.raw & 0xF]

    @property
    def is_query_consuming(self):
        return bool((0x3C1A7 >> ((self.raw & 0xF) * 2)) & 1)

    @property
    def is_reference_consuming(self):
        return bool((0x3C1A7 >> ((self.raw & 0xF) * 2)) & 2)

    def __repr__(self):
        return f"{self.length}{self.type}"


class PileupRead:
    """pileup.d:86-230 over one bgp_entry + the decoded record."""
    __slots__ = ("_col", "_e")

    def __init__(self, col, e):
        self._col, self._e = col, e

    @property
    def _rec(self):
        return self._col._records.rec[int(self._e["read_ordinal"])]

    def _cig(self):
        r = self._col._records
        i = int(self._e["read_ordinal"])
        return r.cigar[int(r.cigar_off[i]):int(r.cigar_off[i + 1])]

    read_ordinal = property(lambda s: int(s._e["read_ordinal"]))
    cigar_operation_offset = property(lambda s: int(s._e["cigar_op_offset"]))
    query_offset = property(lambda s: int(s._e["query_offset"]))
    current_base = property(lambda s: chr(int(s._e["base"])))
    current_base_quality = property(lambda s: int(s._e["qual"]))
    cigar_operation = property(lambda s: CigarOperation(s._cig()[int(s._e["cigar_op_index"])]))
    cigar_before = property(lambda s: [CigarOperation(x) for x in s._cig()[:int(s._e["cigar_op_index"])]])
    cigar_after = property(lambda s: [CigarOperation(x) for x in s._cig()[int(s._e["cigar_op_index"]) + 1:]])
    cigar = property(lambda s: [CigarOperation(x) for x in s._cig()])
    ref_id = property(lambda s: int(s._rec["ref_id"]))
    position = property(lambda s: int(s._rec["pos"]))
    end_position = property(lambda s: int(s._rec["end_pos"]))
    mapping_quality = property(lambda s: (int(s._rec["bin_mq_nl"]) >> 8) & 0xFF)
    flag = property(lambda s: int(s._rec["flag_nc"]) >> 16)
    is_reverse_strand = property(lambda s: bool(s.flag & 0x10))
    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    sequence_length = property(lambda s: int(s._rec["l_seq"]))

    @property
    def sequence(self):
        r = self._col._records
        i = self.read_ordinal
        return bytes(r.seq[int(r.seq_off[i]):int(r.seq_off[i + 1])]).decode("latin-1")

    @property
    def base_qualities(self):
        r = self._col._records
        i = self.read_ordinal
        return r.qual[int(r.seq_off[i]):int(r.seq_off[i + 1])]

    @property
    def name(self):
        data = self._col._data
        if data is None:
            raise AttributeError("read names need the host copy of the record stream")
        off = int(self._rec["rec_off"]) + 36
        ln = int(self._rec["bin_mq_nl"]) & 0xFF
        return data[off:off + ln - 1].decode("latin-1")


class PileupColumn:
    """pileup.d:236-290"""
    __slots__ = ("ref_id", "position", "_c", "_entries", "_records", "_data", "reference_base")

    def __init__(self, ref_id, position, c, entries, records, data):
        self.ref_id, self.position, self._c, self._entries, self._records, self._data = ref_id, position, c, entries, records, data
        self.reference_base = "N"            # pileup.d:245-249 default

    coverage = property(lambda s: int(s._c["coverage"]))
    n_starting_here = property(lambda s: int(s._c["n_starting_here"]))

    @property
    def reads(self) -> List[PileupRead]:
        if self._entries is None:
            raise AttributeError("column was produced without per-read entries (want_entries=False)")
        return [PileupRead(self, e) for e in self._entries]

    @property
    def reads_starting_here(self):
        r = self.reads
        return r[len(r) - self.n_starting_here:]                # pileup.d:272-274

    @property
    def bases(self):
        return bytes(self._entries["base"]).decode("latin-1")   # pileup.d:277-279

    @property
    def base_qualities(self):
        return self._entries["qual"]                             # pileup.d:282-284

    @property
    def read_qualities(self):
        return [r.mapping_quality for r in self.reads]           # pileup.d:287-289

    @property
    def counts(self):
        c = self._c
        return dict(A=int(c["a"]), C=int(c["c"]), G=int(c["g"]), T=int(c["t"]), other=int(c["other"]), deletion=int(c["del"]))


class _Pileup:
    """AbstractPileup (pileup.d:428-458): iterable of columns + start_position/end_position/ref_id."""

    def __init__(self, reads, mode, use_md_tag, start_from, end_at, skip_zero_coverage, want_entries, window_positions):
        self.reads = reads
        self.start_position, self.end_position = start_from, end_at
        self._args = (mode, use_md_tag, start_from, end_at, skip_zero_coverage, want_entries, window_positions)
        self._ctx = reads._stage()
        self.ref_id = -1

    def windows(self) -> Iterator[Window]:
        mode, use_md, s, e, skip, want, win = self._args
        gen = self._ctx.pileup_windows(mode, use_md, s, e, skip, want, win)
        first = True
        for w in gen:
            if first and mode == L.BGP_MAKE_PILEUP:
                self.ref_id = self._ctx.pileup_ref_id()
                first = False
            yield w

    def __iter__(self) -> Iterator[PileupColumn]:
        mode, use_md, s, e, skip, want, win = self._args
        records = None
        if want:
            records = self.reads.records or self.reads.decode()
        for w in self.windows():
            cols = w.columns
            for k in range(len(cols)):
                c = cols[k]
                if skip and c["coverage"] == 0:          # PileupRange.popFront jumps over these, pileup.d:388-392
                    continue
                ents = w.entries[int(w.entry_off[k]):int(w.entry_off[k + 1])] if w.entries is not None else None
                col = PileupColumn(w.ref_id, int(w.first_position) + k, c, ents, records, getattr(self.reads, "data", None))
                if w.reference_base is not None:
                    col.reference_base = chr(int(w.reference_base[k]))     # pileup.d:245-249
                yield col


def makePileup(reads, use_md_tag=False, start_from=0, end_at=ULONG_MAX, skip_zero_coverage=True, *, want_entries=True, window_positions=0):
    """pileup.d:683-694.  Only the first reference of `reads` is processed (pileup.d:490-494)."""
    return _Pileup(reads, L.BGP_MAKE_PILEUP, use_md_tag, start_from, end_at, skip_zero_coverage, want_entries, window_positions)


class _ChunkPileup(_Pileup):
    """One element of PileupChunkRange (pileup.d:859-987): the pileup of a chunk, clipped by the seam rule."""

    def __init__(self, reads, ctx, k, info, use_md_tag, want_entries, window_positions):
        self.reads, self._ctx, self._k = reads, ctx, k
        self.ref_id, self.start_position, self.end_position = int(info.ref_id), int(info.start_position), int(info.end_position)
        self.first_record, self.last_record = int(info.first_record), int(info.last_record)
        self._args = (L.BGP_MAKE_PILEUP, use_md_tag, self.start_position, self.end_position, True, want_entries, window_positions)

    def windows(self) -> Iterator[Window]:
        return self._ctx.chunk_windows(self._k)


def pileupChunks(reads, use_md_tag=False, block_size=16_384_000, start_from=0, end_at=ULONG_MAX, *, want_entries=True, window_positions=0):
    """pileup.d:1011-1015: consecutive non-overlapping pileups (chunks cut as chunksConsumingLessThan does, splitter.d:65-91;
    seams as PileupChunkRange sets them, pileup.d:905-913,932-939).  The elements may be consumed in any order."""
    ctx = reads._stage()
    n = ctx.pileup_chunks_begin(use_md_tag, block_size, start_from, end_at, want_entries, window_positions)
    if want_entries and n and getattr(reads, "records", None) is None:
        reads.decode()
        ctx.pileup_chunks_begin(use_md_tag, block_size, start_from, end_at, want_entries, window_positions)
    return [_ChunkPileup(reads, ctx, k, ctx.pileup_chunk(k), use_md_tag, want_entries, window_positions) for k in range(n)]


def pileupColumns(reads, use_md_tag=False, skip_zero_coverage=True, *, want_entries=True, window_positions=0):
    """pileup.d:509-519."""
    return _Pileup(reads, L.BGP_PILEUP_COLUMNS, use_md_tag, 0, ULONG_MAX, skip_zero_coverage, want_entries, window_positions)


# ------------------------------------------------------------------------------------------ SNP calling (snpcallers/maq.d)

class DiploidCall5:
    """Call!(DiploidGenotype, Base5), bio/core/call.d:29-88."""

    def __init__(self, sample, chromosome, position, reference_base, genotype_code, quality):
        self.sample, self.chromosome, self.position, self.reference_base, self.quality = sample, chromosome, position, reference_base, quality
        self.base1, self.base2 = "ACGTN"[genotype_code // 5], "ACGTN"[genotype_code % 5]     # genotype.d:33-37,51-58

    @property
    def genotype(self) -> str:
        return self.base1 + "|" + self.base2                                                 # genotype.d:71-73

    @property
    def is_heterozygous(self) -> bool:
        return self.base1 != self.base2

    @property
    def is_variant(self) -> bool:
        return not (self.base1 == self.base2 == self.reference_base)                          # call.d:85-87


class MaqSnpCaller:
    """MaqSnpCaller of bio/std/hts/snpcallers/maq.d:308-541: same property names and defaults; the model runs on the
    device (bgp_pileup_maq).  The stated choices where the reference is not deterministic are in include/biodgpu.h."""

    def __init__(self):
        self.depcorr, self.eta, self.minimum_call_quality, self.minimum_base_quality = 0.17, 0.03, 6.0, 13   # maq.d:329-333

    def calls(self, reads, reference="", sample="", use_md_tag=True, window_positions=0):
        """makeCall (maq.d:443-486) for every column with coverage: DiploidCall5 or None (no valid base in the column)."""
        ctx = reads._stage()
        for w in ctx.pileup_windows(L.BGP_PILEUP_COLUMNS, use_md_tag=use_md_tag, window_positions=window_positions,
                                    maq=(self.depcorr, self.eta, self.minimum_base_quality)):
            for k in np.nonzero(w.columns["coverage"])[0]:
                m = w.maq[k]
                if not (int(m["flags"]) & L.BGP_MAQ_CALLED):
                    yield None
                    continue
                yield DiploidCall5(sample, reference, int(w.first_position) + int(k), "ACGTN"[int(m["reference_base5"])], int(m["genotype"]), float(m["quality"]))

    def findSNPs(self, reads, reference="", sample=""):
        """maq.d:488-541: the calls that are variants with quality above minimum_call_quality (needs MD tags: reference bases)."""
        for c in self.calls(reads, reference, sample, use_md_tag=True):
            if c is not None and c.is_variant and c.quality > self.minimum_call_quality:
                yield c
