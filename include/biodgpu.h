/*
 * biodgpu.h -- C ABI of libbiodgpu.so: BioD's BAM-record decode -> CIGAR walk -> pileup hot path on
 * MI355X (gfx950).  This is the drop-in boundary a BioD maintainer binds with `extern (C)`, exactly
 * the way BioD already binds zlib (bio/core/utils/zlib.d:6-245, linked with -L-lz, Makefile:10):
 * plain pointers and sizes, int status codes turned into exceptions by the D caller
 * (cf. bio/core/bgzf/block.d:162-185).  No C++/torch types cross this boundary.
 *
 * Every entry point names the reference interface (file:line in biod/BioD v0.2.4) it replaces.
 * The D-side binding is shown in INTEGRATION.md; a C++ mirror of BioD's range API over this ABI is
 * include/biodgpu_pileup.hpp.
 *
 * Threading (mirrors BioD: the consumer side of the path is single threaded, inflate runs on
 * TaskPool workers -- bio/core/bgzf/inputstream.d:414-428): one bgp_ctx per (host thread, GPU);
 * functions taking a ctx are not re-entrant.  Ownership (mirrors bio/std/hts/bam/pileup.d:396,
 * 660-661 "storage is reused"): every pointer returned through an out-struct is owned by the ctx and
 * stays valid until the next call that produces the same kind of result, or bgp_destroy (the memory behind it stays
 * mapped until bgp_reset / bgp_destroy: a pointer kept too long reads stale bytes, never freed memory -- unless pinned
 * host memory runs out: replaced buffers are then given back, oldest first, before a call answers BGP_ENOMEM).
 *
 * There is NO CPU fallback: without a usable gfx950 device bgp_init fails with BGP_EDEVICE.
 */
#ifndef BIODGPU_H
#define BIODGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGP_ABI_VERSION 4

/* ---- status codes -------------------------------------------------------------------------- */
enum {
    BGP_OK = 0,
    BGP_EINVAL = -1,    /* bad argument; malformed record (sections exceed block_size)              */
    BGP_ETRUNC = -2,    /* record body runs past the data -- readExact, readrange.d:169              */
    BGP_ECIGAR = -3,    /* cursor ran off the CIGAR -- assertCigarIndexIsValid, pileup.d:224-228     */
    BGP_EDEVICE = -4,   /* HIP error / no gfx950 device                                              */
    BGP_ENOMEM = -5,
    BGP_ESTATE = -6,    /* call out of order (e.g. pileup before flush)                              */
    BGP_EMD = -7,       /* MD tag missing/short (use_md_tag)                                         */
    BGP_EUNSORTED = -8, /* reads not coordinate sorted -- precondition of makePileup, pileup.d:25-27 */
    BGP_ESEQ = -9,      /* soft: a query offset ran past l_seq (RangeError in read.d:364-383);
                           results are still produced, such bases counted as "other"                 */
    BGP_EOF = 1,        /* iteration finished                                                        */
    BGP_EAGAIN = 2      /* batched staging: push more input / hand out windows first                 */
};

typedef struct bgp_ctx bgp_ctx;

const char* bgp_strerror(int status);
int bgp_abi_version(void);
/* Human-readable detail of the last failure on this ctx (or of the last HIP failure when ctx is NULL). */
const char* bgp_last_error(bgp_ctx* ctx);

/* Creates a context on HIP device `device`.  Replaces nothing in BioD (BioD has no device); plays
 * the role of `new BamReader(fn, taskPool)` resource setup, bio/std/hts/bam/reader.d:127-138. */
int bgp_init(int device, bgp_ctx** out);
void bgp_destroy(bgp_ctx* ctx);
/* Device memory the library holds (all contexts of the process), now and at its peak; reset_peak != 0 makes the peak start over.
 * bgp_set_device_memory_cap: an allocation that would take the total beyond `bytes` fails with BGP_ENOMEM (0: no cap; the environment
 * variable BGP_HBM_CAP sets it at start) -- the bounded-memory file path (bgp_stream_bam_begin) is tested under one. */
int bgp_device_memory(uint64_t* current, uint64_t* peak, int reset_peak);
int bgp_set_device_memory_cap(uint64_t bytes);
/* Use a caller-owned hipStream_t for every launch (NULL = the ctx's own stream). */
int bgp_set_stream(bgp_ctx* ctx, void* hip_stream);
/* Number of reference sequences from the BAM header (tightens the record-boundary guesser; 0 = unknown). */
int bgp_set_n_ref(bgp_ctx* ctx, int32_t n_ref);
/* Drops all staged data and results; keeps allocations for reuse. */
int bgp_reset(bgp_ctx* ctx);

/* ---- staging: inflated record bytes ---------------------------------------------------------
 * Replaces BgzfInputStream.readBlock feeding BamReadRange (inputstream.d:497-530, readrange.d:137-169):
 * the caller hands over already-inflated BGZF payloads in file order, starting at the first record
 * (after the BAM header, reader.d:121-123).  `coffset` is the block's compressed file offset (kept
 * for virtual offsets, virtualoffset.d:29-59).  `first_record_uoffset` is a hint: offset inside this
 * block of the first record that STARTS in it, or BGP_UOFFSET_UNKNOWN.  The hint is never trusted: the boundary
 * guesser tries the hinted offset first, by the same plausibility test it applies to its own candidates, and every
 * boundary is verified by the chain of records on the device -- a wrong hint costs time, never correctness. */
#define BGP_UOFFSET_UNKNOWN 0xFFFFFFFFu
int bgp_push_inflated(bgp_ctx* ctx, const uint8_t* data, size_t len, uint64_t coffset, uint32_t first_record_uoffset);
/* Zero-copy variant: the record stream is already resident in HBM (device pointer, caller keeps it
 * alive until bgp_reset/bgp_destroy). */
int bgp_attach_device_stream(bgp_ctx* ctx, const void* device_ptr, uint64_t len);

/* Convenience file path: BGZF inflate on host worker threads (the TaskPool analogue,
 * inputstream.d:386-483, block.d:127-216), BAM magic + header + reference table
 * (reader.d:111-116,579-597), then pushes every record block.  n_threads <= 0: all cores. */
int bgp_load_bam(bgp_ctx* ctx, const char* path, int n_threads);
/* The same file path with decompressBgzfBlock (block.d:127-216: raw DEFLATE, size check, CRC32 check) on the
 * DEVICE: the file is read into HBM compressed, one wave inflates one BGZF block, the record stream never exists
 * on the host (bgp_records.stream is NULL unless BGP_FETCH_STREAM asks for a copy).  For hosts with few cores per
 * GPU.  Same errors as bgp_load_bam; a block that fails to inflate or fails its CRC is BGP_EINVAL. */
int bgp_load_bam_device(bgp_ctx* ctx, const char* path);
int bgp_n_ref(bgp_ctx* ctx, int32_t* n_ref);
int bgp_ref_info(bgp_ctx* ctx, int32_t ref_id, const char** name, int32_t* length);   /* referenceinfo.d */

/* ---- decode -----------------------------------------------------------------------------------
 * bgp_flush = kernels K0+K1: record framing (readrange.d:118-173) and the fixed fields / offsets /
 * end_position of every record (read.d:907-1003, 255-262, 1380-1383). */
typedef struct bgp_record {      /* 32 bytes per record, SURVEY.md 8(a2) */
    int32_t ref_id;              /* read.d:907  _refID                              */
    int32_t pos;                 /* read.d:911  _pos                                */
    int32_t end_pos;             /* read.d:1382 position + basesCovered()           */
    uint32_t bin_mq_nl;          /* read.d:915  bin<<16 | mapq<<8 | l_read_name     */
    uint32_t flag_nc;            /* read.d:919  flag<<16 | n_cigar_op               */
    int32_t l_seq;               /* read.d:923                                      */
    uint64_t rec_off;            /* offset of the record's block_size prefix in the stream */
} bgp_record;

int bgp_flush(bgp_ctx* ctx);
/* bgp_flush with the arenas of `what` (BGP_FETCH_SEQ | _QUAL | _CIGAR) made in the same pass over the stream: framing,
 * rows and unpack are one kernel that reads every stream byte once (the shape of BamReadRange feeding
 * read.sequence / base_qualities / cigar as it goes, readrange.d:118-173 + read.d:225-228,364-383,468-470).  A
 * following bgp_decode_device / _fetch for the same sections finds them done.  Same results and errors as bgp_flush
 * followed by bgp_decode_device(what).  `what | BGP_DECODE_ONLY`: the caller's word that no pileup will follow -- the pass then
 * writes nothing for the pileup kernels (a pileup asked for anyway is correct, it plans every read itself). */
int bgp_flush_decode(bgp_ctx* ctx, uint32_t what);
int bgp_num_records(bgp_ctx* ctx, uint64_t* n);

typedef struct bgp_records {
    uint64_t n;
    const bgp_record* rec;       /* n                                                            */
    const uint64_t* seq_off;     /* n+1: exclusive prefix sum of l_seq      (seq and qual arenas) */
    const uint64_t* cigar_off;   /* n+1: exclusive prefix sum of n_cigar_op                      */
    const uint8_t* seq;          /* BamRead.sequence as chars, read.d:364-383,438-440, base.d:85 */
    const uint8_t* qual;         /* BamRead.base_qualities, read.d:468-470                       */
    const uint32_t* cigar;       /* BamRead.cigar raw ops, read.d:225-228                        */
    const uint8_t* stream;       /* host copy of the raw record stream (names, tags), or NULL    */
    uint64_t stream_len;
} bgp_records;

#define BGP_FETCH_SEQ 1u
#define BGP_FETCH_QUAL 2u
#define BGP_FETCH_CIGAR 4u
#define BGP_FETCH_ALL 7u
#define BGP_FETCH_STREAM 8u   /* bgp_decode_fetch: also a host copy of the record stream when the loader kept none */
#define BGP_DECODE_ONLY 16u   /* bgp_flush_decode: no pileup is going to follow -- the pass writes the record table and the arenas and nothing for K4
                              * (its plan rows: 17 bytes a record).  A pileup asked for all the same is correct, it plans every read itself. */
/* Kernel K2 (4-bit seq -> chars, qual copy, CIGAR copy) + copy to host memory owned by the ctx. */
int bgp_decode_fetch(bgp_ctx* ctx, uint32_t what, bgp_records* out);
/* Same products left in HBM (device pointers); nothing is copied to the host. */
int bgp_decode_device(bgp_ctx* ctx, uint32_t what, bgp_records* out_device);
/* bgp_decode_fetch for the records [first, first + count) only -- a slice of BamReader.reads (BioD's records are slices
 * into the inflated blocks, readrange.d:176-187; a BamRead never owns its bytes, read.d:876-883).  out->n = count;
 * rec has count rows (rec_off stays the offset in the WHOLE stream); seq_off / cigar_off have count + 1 entries and keep
 * their stream-wide values, so record first + i owns seq[seq_off[i] - seq_off[0] .. seq_off[i + 1] - seq_off[0]) of the
 * returned arena slice (same for qual; cigar with cigar_off).  With BGP_FETCH_STREAM out->stream is a host copy of the
 * stream bytes from rec_off of record `first` up to the record behind the slice (or the end of the stream), so record
 * first + i begins at stream + (rec[i].rec_off - rec[0].rec_off): read names (read.d:218-222) and tags (read.d:1070-1087)
 * of the reads of a window of columns without copying a stream that may not fit host memory.  first + count beyond the
 * table is BGP_EINVAL. */
int bgp_decode_fetch_range(bgp_ctx* ctx, uint32_t what, uint64_t first, uint64_t count, bgp_records* out);

/* ---- pileup -------------------------------------------------------------------------------- */
typedef struct bgp_column {      /* per reference position, 32 bytes (SURVEY.md 8 a-out summary) */
    uint32_t coverage;           /* PileupColumn.coverage, pileup.d:252                          */
    uint32_t n_starting_here;    /* _n_starting_here, pileup.d:241,272-274                       */
    uint32_t a, c, g, t;         /* counts of current_base (pileup.d:115-122) over the column     */
    uint32_t other;              /* any other M/=/X base (N, IUPAC, '=')                          */
    uint32_t del;                /* '-' : current op is D or N                                    */
} bgp_column;

typedef struct bgp_entry {       /* one (read, column) pair, 16 bytes (SURVEY.md 8 a-out entries) */
    uint32_t read_ordinal;       /* index of the record in the stream                             */
    uint32_t cigar_op_offset;    /* PileupRead.cigar_operation_offset, pileup.d:99                */
    uint32_t query_offset;       /* PileupRead.query_offset, pileup.d:146                         */
    uint16_t cigar_op_index;     /* index of cigar_operation; cigar_before/after split, pileup.d:104-111 */
    uint8_t base;                /* current_base, pileup.d:115-122                                */
    uint8_t qual;                /* current_base_quality, pileup.d:127-134                        */
} bgp_entry;

/* A run of the sweep: a reference's reads from one stretch without coverage to the next.  A reference is ONE run unless
 * 65 536 or more of its positions in a row are uncovered: BioD's sweep jumps over such a stretch (skip_zero_coverage,
 * initNewReference, pileup.d:388-392,399-424) or walks it column by column; here it simply has no columns on the device (two
 * reads 2^31 positions apart cost two windows, not 64 GB).  Runs are in file order; runs of one reference follow each other.
 * The window iterators hand the stretch out as windows of zero columns (reference_base 'N') when skip_zero_coverage is 0, and
 * step over it otherwise; bgp_batch_next_window does not hand such stretches out. */
typedef struct bgp_run {
    int32_t ref_id;
    uint32_t _pad;
    uint64_t first_record, end_record;  /* record index range holding the run's reads            */
    uint64_t start_pos, end_pos;        /* columns exist for start_pos <= position < end_pos      */
    uint64_t column_base;               /* index of start_pos in the dense column array           */
} bgp_run;

typedef struct bgp_pileup_info {
    uint32_t n_runs;
    const bgp_run* runs;         /* host, ctx owned                                              */
    uint64_t n_positions;        /* dense positions over all runs                                */
    uint64_t n_columns;          /* positions with coverage > 0 (= columns BioD yields with skip_zero_coverage) */
    uint64_t sum_coverage;
    uint32_t max_coverage;
    int32_t soft_status;         /* BGP_OK or BGP_ESEQ                                           */
    uint64_t n_reads_used;       /* reads with basesCovered() > 0, pileup.d:481,510              */
    const bgp_column* device_columns; /* dense array in HBM, n_positions entries                 */
    const uint64_t* device_depth_hist; /* HBM, BGP_DEPTH_BINS bins: #positions with coverage d (clipped) */
} bgp_pileup_info;
#define BGP_DEPTH_BINS 65536

/* Kernels K3+K4 over every staged read: the whole-stream equivalent of
 * pileupColumns(reads, false, skip_zero_coverage) (pileup.d:509-519).  Requires bgp_flush. */
int bgp_pileup_run(bgp_ctx* ctx, bgp_pileup_info* info);

/* Column iteration with BioD's argument lists.
 *   mode BGP_MAKE_PILEUP    = makePileup(reads, use_md_tag, start_from, end_at, skip_zero_coverage), pileup.d:683-694
 *   mode BGP_PILEUP_COLUMNS = pileupColumns(reads, use_md_tag, skip_zero_coverage),                   pileup.d:509-519
 * next_window hands out consecutive dense windows (zero-coverage positions included -- the D/C++
 * facade skips them when skip_zero_coverage is set, exactly the columns PileupRange.popFront
 * pileup.d:345-397 would have yielded).
 *
 * use_md_tag != 0 selects PileupRangeUsingMdTag (pileup.d:522-654): every window also carries
 * reference_base, the base BioD's column.reference_base (pileup.d:245-249) reports -- rebuilt from the MD
 * tags by the same provider chain (add() pileup.d:563-596, chunk refill :627-652), including its quirks
 * ('N' while no chunk is available; chunk_end_position carried over from the previous reference).
 * Where BioD's popFront would throw on an MD tag that disagrees with its read, bgp_pileup_begin returns
 * BGP_EMD (or BGP_EINVAL for malformed aux data) -- for BGP_MAKE_PILEUP only if that popFront lies before
 * end_at, as BioD stops popping there (pileup.d:497-505).  With skip_zero_coverage the bytes at
 * zero-coverage positions (columns BioD never yields) are unspecified. */
#define BGP_PILEUP_COLUMNS 0
#define BGP_MAKE_PILEUP 1
typedef struct bgp_columns {
    int32_t ref_id;
    uint32_t n_positions;
    uint64_t first_position;
    const bgp_column* columns;   /* host, n_positions                                             */
    const uint64_t* entry_off;   /* host, n_positions+1, or NULL when entries were not requested  */
    const bgp_entry* entries;    /* host, reads of column k: entries[entry_off[k] .. entry_off[k+1]) in file order */
    const uint8_t* reference_base; /* host, n_positions (use_md_tag) or NULL                      */
    const uint8_t* snp;          /* host, n_positions of BGP_SNP_* after bgp_pileup_snp, else NULL */
    const struct bgp_maq_call* maq; /* host, n_positions after bgp_pileup_maq, else NULL             */
    const struct bgp_column8* columns8; /* host, n_positions, INSTEAD of columns (then NULL) under bgp_set_compact_columns */
} bgp_columns;
/* Compact transport of the windows: the fields of bgp_column, one byte each.  With bgp_set_compact_columns(ctx, 1) every window whose
 * counts all fit a byte (coverage <= 255 at every position) is handed out as columns8 -- 8 bytes a position over the link instead of
 * 32 -- and bgp_columns.columns is NULL; a window with a deeper position comes in the 32-byte layout as always (columns8 NULL).  The
 * consumer sees the same numbers (PileupColumn's accessors, pileup.d:236-290); the dense array on the device (bgp_pileup_info
 * .device_columns) is untouched.  The setting holds for the ctx until it is changed (bgp_reset keeps it); default 0. */
typedef struct bgp_column8 { uint8_t coverage, n_starting_here, a, c, g, t, other, del; } bgp_column8;
int bgp_set_compact_columns(bgp_ctx* ctx, int on);
/* makePileup over a FILTERED range of reads.  makePileup / pileupColumns take any input range of reads (pileup.d:683-687, read.d:1402-1420), and
 * callers hand them reads.filter!(r => ...) over the flag properties of read.d:137-194 (is_duplicate 0x400, failed_quality_control 0x200,
 * is_secondary_alignment 0x100, is_unmapped 0x4, proper_pair 0x2 ...) and mapping_quality.  With a filter set, a read passes when
 *     (flag & flag_require) == flag_require  &&  (flag & flag_exclude) == 0  &&  mapping_quality >= min_mapping_quality
 * and a read that does not is not in the range the pileup sees: it takes part in no column, holds no entry, provides no reference base --
 * exactly the reads pileupInstance drops itself for covering nothing (pileup.d:481,510).  The filter is applied where bgp_flush /
 * bgp_flush_decode (and every batch of a batched or file-fed pass) makes the record table: set it BEFORE staging is flushed; it holds for the
 * ctx until changed (0, 0, 0 = none; bgp_reset keeps it).  Read ordinals (bgp_entry.read_ordinal) stay those of the unfiltered stream, and so
 * do pileupChunks' cut points; bgp_records.rec[i].end_pos of a dropped read equals its pos; bgp_bai_build answers BGP_ESTATE under a filter.
 * Any other predicate is the caller's: BioD's CPU templates over the decoded reads. */
int bgp_set_read_filter(bgp_ctx* ctx, uint32_t flag_require, uint32_t flag_exclude, uint32_t min_mapping_quality);
int bgp_pileup_begin(bgp_ctx* ctx, int mode, int use_md_tag, uint64_t start_from, uint64_t end_at,
                     int skip_zero_coverage, int want_entries, uint32_t window_positions);
/* makePileup(bam[ref_id].reads, use_md_tag, start_from, end_at, skip_zero_coverage): bgp_pileup_begin(BGP_MAKE_PILEUP)
 * over the reads of ONE reference of the staged stream, whichever it is -- what a BioD caller writes as
 * makePileup(bam.reference(ref_id)[], ...) (reference.d:84-86 feeding pileup.d:683-694); makePileup itself only ever
 * sees the first reference of its input (pileup.d:490-494).  The provider chain of use_md_tag starts afresh at that
 * reference, as it does for a range that begins there.  A reference without covering reads gives BGP_EOF at the first
 * window. */
int bgp_pileup_begin_ref(bgp_ctx* ctx, int32_t ref_id, int use_md_tag, uint64_t start_from, uint64_t end_at,
                         int skip_zero_coverage, int want_entries, uint32_t window_positions);
int bgp_pileup_next_window(bgp_ctx* ctx, bgp_columns* out);   /* BGP_OK, BGP_EOF, or error */
int bgp_pileup_ref_id(bgp_ctx* ctx, int32_t* ref_id);          /* AbstractPileup.ref_id, pileup.d:455 */

/* ---- batched staging: streams larger than HBM, or arriving over time ---------------------------------------------
 * The shape of BgzfInputStream's bounded ring of blocks in flight (inputstream.d:386-447) under pileupColumns(reads)
 * (pileup.d:509-519; use_md_tag = false): push inflated record bytes in file order with any cut points, take dense
 * windows of FINAL columns out as they become final.  A batch (batch_bytes of input, 0 = 256 MiB) is copied to HBM
 * while the previous one is still in its kernels (three pinned host buffers -- one taking the caller's bytes, full ones queued --,
 * two device buffers, a copy stream of its own: a full buffer leaves for the device the moment a device buffer is free, at the
 * push / commit that filled it or inside the bgp_batch_next_window that moved on to the next batch); the reads that still overlap the position of a batch's last covering read -- found exactly, with the prefix
 * maxima of K3 -- and a partial last record are carried over to the next batch on the device.  Columns and entries
 * (read_ordinal counts over the whole stream) are identical to staging everything at once.
 *   bgp_batch_push         BGP_OK (all bytes taken) or BGP_EAGAIN with *taken < len: every pinned buffer is full and both
 *                          device buffers are taken -- take a window (bgp_batch_next_window), push the rest.  Taking ONE window
 *                          and offering bytes again, in turn, keeps both directions of the link busy
 *   bgp_batch_next_window  BGP_OK (a window), BGP_EAGAIN (push more or bgp_batch_end), BGP_EOF (after bgp_batch_end)
 * Between windows the ctx describes the CURRENT batch: bgp_decode_fetch etc. see its records; bgp_batch_ordinal_base
 * gives the stream-wide ordinal of its record 0.  bgp_reset leaves the mode.  bgp_set_n_ref before bgp_batch_begin holds
 * for every batch.  With want_entries a stream of more than 2^32 records is BGP_EINVAL at the window whose entries would
 * no longer fit bgp_entry.read_ordinal (columns alone have no such bound).  A record whose block_size cannot belong to a
 * BAM record (beyond 2^30, or smaller than the sections its header announces) is BGP_EINVAL in the batch it appears in,
 * not "the rest is still to come". */
int bgp_batch_begin(bgp_ctx* ctx, uint64_t batch_bytes, int want_entries, uint32_t window_positions);
/* bgp_batch_begin FED BY A FILE, in bounded device memory: new BamReader(filename) under pileupColumns(reads) as BioD runs it -- a ring of
 * BGZF blocks in flight (read, inflated by decompressBgzfBlock on a worker, consumed in file order: bio/core/bgzf/inputstream.d:386-447,
 * 455-483, block.d:127-216) under a consumer that pulls, the file never held whole.  Reader threads bring the file to the device
 * compressed, 8 MiB at a time, into a ring; k_inflate (a wave a BGZF block, size and CRC32 checked as block.d:197-207) inflates a batch of
 * at most batch_bytes (0 = 256 MiB; at least 128 KiB) from the ring straight into one of the pass's two device buffers while the previous
 * batch is in its kernels; no inflated byte exists on the host.  The BAM header is read here (bgp_n_ref / bgp_ref_info answer afterwards);
 * then bgp_batch_next_window hands out the windows until BGP_EOF -- it never answers BGP_EAGAIN: nobody pushes (bgp_batch_push / _acquire /
 * _commit / _end are BGP_ESTATE).  bgp_batch_set_md, bgp_pileup_maq, bgp_batch_ordinal_base, bgp_decode_fetch between windows: as for any
 * batched pass.  Device memory is two batch buffers, the ring (batch_bytes and a few chunks) and ONE batch's tables and columns, whatever
 * the file's size (bgp_device_memory).  Errors: those of bgp_load_bam_device, from the bgp_batch_next_window whose batch holds the block. */
int bgp_stream_bam_begin(bgp_ctx* ctx, const char* path, uint64_t batch_bytes, int want_entries, uint32_t window_positions);
/* use_md_tag for the batches (call right after bgp_batch_begin): every window carries reference_base as PileupRangeUsingMdTag
 * (pileup.d:522-654) yields it over the WHOLE stream -- the provider chain's state (provider, cursor, chunk_end_position, the next
 * read add() has not seen) is carried across the seam between two batches, so examples/make_pileup.d:16, makePileup(bam.reads,
 * true), runs on a file that does not fit the device.  skip_zero_coverage says which sweep the chain belongs to (it re-initialises
 * at uncovered columns only then, pileup.d:388-392).  BGP_EMD comes from the bgp_batch_next_window whose batch holds the column
 * BioD's popFront would throw at. */
int bgp_batch_set_md(bgp_ctx* ctx, int use_md_tag, int skip_zero_coverage);
int bgp_batch_push(bgp_ctx* ctx, const uint8_t* data, size_t len, size_t* taken);
/* bgp_batch_push for bytes that already lie in PAGE-LOCKED host memory (hipHostMalloc, hipHostRegister -- where an inflater that owns its buffers
 * keeps them): they are copied to the device from where they are, batch_bytes a device buffer, and never into the pass's own pinned buffers.
 * ALL of `len` is taken (*taken = len): what no device buffer is free for yet waits where it lies and leaves as buffers fall free -- inside the
 * bgp_batch_next_window that moves on to the next batch, so the link never waits for the caller.  The bytes must stay as they are until the
 * windows of their batches have been handed out or the pass is over.  While bytes of an earlier call still wait, a further call is taken only
 * if its bytes lie right behind them (else BGP_EAGAIN: take a window first), and bgp_batch_push / _acquire answer BGP_EAGAIN: what was pushed
 * first is piled up first.  Bytes that are not page-locked are refused (BGP_EINVAL). */
int bgp_batch_push_pinned(bgp_ctx* ctx, const uint8_t* data, size_t len, size_t* taken);
/* bgp_batch_push without the copy: bgp_batch_acquire hands out the free tail of the pinned buffer the next batch is copied to HBM
 * from (*cap bytes at *ptr; BGP_EAGAIN: hand out windows first), the caller -- an inflater, as decompressBgzfBlock writes into the
 * block's own buffer, block.d:127-216 -- fills it, bgp_batch_commit says how many bytes it wrote (<= *cap; 0 is fine).  Pushes and
 * acquire / commit pairs may alternate, but between an acquire and its commit the buffer is the caller's: bgp_batch_push, a second
 * bgp_batch_acquire and bgp_batch_end answer BGP_ESTATE until it is committed.  A tail shorter than one inflated BGZF block (64 KiB)
 * is not handed out: the buffer is shipped as it is and the next one's beginning comes back. */
int bgp_batch_acquire(bgp_ctx* ctx, uint8_t** ptr, size_t* cap);
int bgp_batch_commit(bgp_ctx* ctx, size_t n);
int bgp_batch_end(bgp_ctx* ctx);
int bgp_batch_next_window(bgp_ctx* ctx, bgp_columns* out);
int bgp_batch_ordinal_base(bgp_ctx* ctx, uint64_t* base);

/* pileupChunks(reads, use_md_tag, block_size = 16_384_000, start_from = 0, end_at = ulong.max), pileup.d:1011-1015:
 * consecutive non-overlapping pileups a caller maps over in parallel.  The chunks are cut exactly as
 * chunksConsumingLessThan does (splitter.d:65-91: size_in_bytes of the records, never across a ref_id), walked as
 * PileupChunkRange does (pileup.d:859-939: chunks of unmapped reads skipped, leading chunks ending at or before
 * start_from dropped, empty when the first kept chunk begins at or after end_at), and chunk j's columns are those of
 * makePileup(..., start_position, end_position) with BioD's seam rule: start_position = position of the previous
 * chunk's last read (same reference) or of the chunk's own first read, end_position = position of the chunk's last
 * read, or the right end when the reference ends with the chunk -- each clipped by start_from / end_at.
 * Two deviations, stated: every read overlapping a chunk's columns is counted (BioD keeps the previous chunk's reads
 * that start within twice a SAMPLED median read length, pileup.d:944-985), and the last chunk of a reference ends at
 * the largest end_position of the reference, not of that chunk; with them the chunks' columns concatenate to exactly
 * pileupColumns'.  With use_md_tag the reference bases are those of the whole-stream provider chain (a fresh chain per
 * chunk, as BioD runs it, can differ where MD tags disagree with each other).
 * bgp_pileup_chunks_begin plans the chunks (*n_chunks of them); bgp_pileup_chunk selects chunk k (any order, any number
 * of times): its description goes to *out and bgp_pileup_next_window then hands out its windows until BGP_EOF. */
typedef struct bgp_chunk_info {
    int32_t ref_id; uint32_t _pad;
    uint64_t first_record, last_record;     /* the chunk's own reads (ordinals in the record table), inclusive */
    uint64_t start_position, end_position;  /* AbstractPileup.start_position / end_position, pileup.d:439-452 */
} bgp_chunk_info;
int bgp_pileup_chunks_begin(bgp_ctx* ctx, int use_md_tag, uint64_t block_size, uint64_t start_from, uint64_t end_at,
                            int want_entries, uint32_t window_positions, uint64_t* n_chunks);
int bgp_pileup_chunk(bgp_ctx* ctx, uint64_t k, bgp_chunk_info* out);

/* isSNP(column, settings) of bio/std/hts/snpcallers/simple.d:15-41 for every position of the windows that
 * follow (call after bgp_pileup_begin with use_md_tag = 1; defaults of SimpleCallerSettings: 5, 2).
 * coverage < minimum_coverage -> no; no base other than '-' -> no; consensus = the base with the largest
 * count; fewer than minimum_witnesses of it -> no; else consensus != reference_base.
 * BioD takes the consensus with argmax over `int[char].byKey` (bio/core/utils/algo.d:79-93): the FIRST largest
 * count in druntime's hash-table iteration order, which nothing in the reference pins.  Where that order decides
 * the answer (several bases share the largest count and reference_base is one of them) the result is BGP_SNP_TIE
 * instead of a guess.  The column summary merges the IUPAC ambiguity codes and '=' into one counter; where they
 * could hold the largest count the result is BGP_SNP_NEEDS_ENTRIES and the caller decides from the per-read
 * entries (the C++ / Python facades do). */
#define BGP_SNP_NO 0
#define BGP_SNP_YES 1
#define BGP_SNP_TIE 2
#define BGP_SNP_NEEDS_ENTRIES 3
int bgp_pileup_snp(bgp_ctx* ctx, int minimum_coverage, int minimum_witnesses);

/* MaqSnpCaller.makeCall (bio/std/hts/snpcallers/maq.d:443-486) for every position of the windows that follow (call after
 * bgp_pileup_begin -- or after bgp_batch_begin, for the windows of every batch; with use_md_tag the call knows its
 * reference base, without it -- and in batches -- the base is 'N' as in :455-459):
 * the reads of a column that count (current_base_quality >= minimum_base_quality, current_base != '-', :400-412), each
 * with quality min(base quality, mapping quality) and its strand; computeLikelihoods (:133-240) over them with the
 * error model ErrorModelCoefficients(depcorr, eta) (:73-118; MaqSnpCaller's defaults 0.17, 0.03, minimum base quality
 * 13); the genotype with the lowest score and the distance to the second lowest (:245-279,466-471).
 * findSNPs (:488-541) is then: (flags & BGP_MAQ_CALLED) && (flags & BGP_MAQ_VARIANT) && quality > minimum_call_quality
 * (default 6.0) -- a filter the caller applies.
 * Where the reference's result is not a function of its input, stated: its sort by quality is unstable and its sums
 * depend on the order among equal qualities (:146-166) -- the order here is file order; above 255 valid bases it takes
 * a RANDOM sample of 255 (:141-143) -- here the first 255, flagged BGP_MAQ_SUBSAMPLED; it subtracts in 80-bit `real`
 * before rounding a heterozygous score to float (:203-207) -- done in double here (one float ulp at most).  The reference
 * has no test for this module: results are identical to oracle/maq_oracle.py, which restates it with the same choices. */
typedef struct bgp_maq_call {
    float quality;               /* gts[1].score - gts[0].score, maq.d:470 (call.quality)                          */
    uint8_t genotype;            /* DiploidGenotype!Base5 code of gts[0]: base1 * 5 + base2, A C G T N = 0..4 (genotype.d:33-37); 0xFF: no call */
    uint8_t reference_base5;     /* Base5(column.reference_base): 0..3, 4 = N                                      */
    uint8_t flags;               /* BGP_MAQ_*                                                                      */
    uint8_t n_bases;             /* bases that entered computeLikelihoods (<= 255)                                 */
} bgp_maq_call;
#define BGP_MAQ_CALLED 1u        /* the column has a valid base: makeCall returns a call                            */
#define BGP_MAQ_VARIANT 2u       /* call.is_variant: genotype != homozygous reference (call.d:85-87)                */
#define BGP_MAQ_SUBSAMPLED 4u    /* more than 255 valid bases: the first 255 were used                              */
int bgp_pileup_maq(bgp_ctx* ctx, float depcorr, float eta, uint32_t minimum_base_quality);
/* ErrorModelCoefficients(depcorr, eta) (maq.d:73-118) as the device uses them, computed on the host: fk[256],
 * lhet[256 * 256] (index n << 8 | k), beta[64 * 256 * 256] (index q << 16 | n << 8 | k).  Needs no device. */
int bgp_maq_error_model(float depcorr, float eta, double* fk, double* lhet, double* beta);

/* dna(read): the reference bases a read spans, rebuilt from its CIGAR and MD tag
 * (bio/std/hts/bam/md/reconstruct.d:38-214, mdOperations bio/std/hts/bam/md/parse.d:13-142).
 * Writes at most `cap` chars to `out`; *len is the full length.  Returns BGP_EMD when the lazy range
 * BioD builds would run into an exhausted query (MD longer than the CIGAR's M/=/X bases). */
int bgp_read_dna(bgp_ctx* ctx, uint64_t record, char* out, uint32_t cap, uint32_t* len);

/* ---- BAI index and region queries: bam["chr"][beg .. end] feeding the device path (host side) ----
 * Virtual offsets are coffset << 16 | uoffset (bio/core/bgzf/virtualoffset.d:30-52); a chunk is a pair of
 * them (bio/core/bgzf/chunk.d:28-40). */
typedef struct bgp_bai bgp_bai;
typedef struct bgp_chunk { uint64_t beg, end; } bgp_chunk;
/* BaiFile(filename): accepts x.bam (tries x.bam.bai, then x.bai) or the .bai itself; parses section 4.2
 * (bio/std/hts/bam/baifile.d:92-169). */
int bgp_bai_open(const char* bam_or_bai_path, bgp_bai** out);
void bgp_bai_close(bgp_bai* bai);
int bgp_bai_n_ref(const bgp_bai* bai, int32_t* n_ref);
int bgp_bai_shape(const bgp_bai* bai, int32_t ref_id, uint32_t* n_bins, uint32_t* n_chunks, uint32_t* n_intervals);
/* Index.getMinimumOffset (baifile.d:77-82): linear-index lower bound for reads overlapping `position`. */
int bgp_bai_min_offset(const bgp_bai* bai, int32_t ref_id, int32_t position, uint64_t* voffset);
/* RandomAccessManager.getChunks (randomaccessmanager.d:209-244): bins that can overlap [beg, end), their
 * chunks ending after the linear-index offset (clamped to it), sorted, overlaps merged (algo.d:95-147).
 * The array is owned by `bai` and valid until the next call. */
int bgp_bai_chunks(bgp_bai* bai, int32_t ref_id, uint32_t beg, uint32_t end, const bgp_chunk** chunks, uint32_t* n);
int bgp_reg2bin(int32_t beg, int32_t end);                       /* bai/bin.d:82-92 */
int bgp_bin_can_overlap(uint32_t bin_id, int32_t beg, int32_t end);   /* Bin.canOverlapWith, bai/bin.d:55-77 */
/* The inflated bytes a chunk list stands for (StreamChunksSupplier, bio/core/bgzf/inputstream.d:254-330): every
 * block from beg.coffset to end.coffset, minus beg.uoffset leading bytes of the first and everything from
 * end.uoffset on in the last.  *len is always set; bytes are written only when cap >= *len. */
int bgp_bgzf_read_chunks(const char* bam_path, const bgp_chunk* chunks, uint32_t n, int n_threads,
                         uint8_t* out, uint64_t cap, uint64_t* len);
/* bam[ref_id][beg .. end] (reference.d:73-81 -> getReads, randomaccessmanager.d:330-336): reads the BAM
 * header, stages the records of getChunks(region); the following bgp_flush frames them and keeps exactly
 * the reads BamReadFilter (randomaccessmanager.d:366-462) yields.  beg < end is enforced (reference.d:76). */
int bgp_load_bam_region(bgp_ctx* ctx, const char* bam_path, bgp_bai* bai, int32_t ref_id, uint32_t beg, uint32_t end, int n_threads);
/* The same with decompressBgzfBlock (block.d:127-216) on the DEVICE, as bgp_load_bam_device does it for a whole file: the chunks' blocks are
 * read and uploaded compressed, a wave inflates a block (size and CRC32 checked), the blocks a chunk cuts (beg.uoffset, end.uoffset:
 * inputstream.d:281-324) are trimmed on the device.  No zlib on the host's cores -- eight ranks a node reading their shards of one file
 * share them.  Same staged bytes, same records, same errors as bgp_load_bam_region. */
int bgp_load_bam_region_device(bgp_ctx* ctx, const char* bam_path, bgp_bai* bai, int32_t ref_id, uint32_t beg, uint32_t end);
/* BamReadFilter applied to the flushed record table in place (any staged stream, e.g. a whole file). */
int bgp_filter_region(bgp_ctx* ctx, int32_t ref_id, uint32_t beg, uint32_t end);
/* getReadAt (randomaccessmanager.d:156-164): stages the single record that starts at `voffset`. */
int bgp_load_bam_read_at(bgp_ctx* ctx, const char* bam_path, uint64_t voffset);
/* BamReadBlock.start_virtual_offset of record i (reads!withOffsets, readrange.d:64-86); needs a stream
 * staged by bgp_load_bam / bgp_load_bam_region / bgp_load_bam_read_at. */
int bgp_record_voffset(bgp_ctx* ctx, uint64_t record, uint64_t* voffset);

/* createIndex (bio/std/hts/bam/bai/indexing.d:353-366; BamReader.createIndex, reader.d:158-164): the .bai of
 * the file staged by ONE bgp_load_bam, after bgp_flush.  Evaluates IndexBuilder.put/finish (indexing.d:56-350,
 * check_bins off) over the device record table: bin runs -> chunks with the same-block merge rule (:221-246),
 * the 16 kb linear index with holes filled from the left (:134-185), the 37450 pseudo-bin (:197-202) and
 * n_no_coor (:347).  Bins are written in ascending id order (the reference's order is druntime's hash order,
 * :193).  *bytes is owned by ctx and valid until the next call.  BGP_EUNSORTED = the builder's "not
 * coordinate-sorted" enforce (:253-268); BGP_EINVAL = a reference id beyond the header's table or a position
 * beyond the linear index (a RangeError in the reference); BGP_ESTATE = anything but one whole file staged. */
int bgp_bai_build(bgp_ctx* ctx, const uint8_t** bytes, uint64_t* n_bytes);

/* Whole-BAM histogram reduce across ranks (the only collective of the path): sums `n` uint64
 * values in HBM over all ranks of an RCCL communicator (ncclComm_t passed as void*).
 * CPU equivalent: a fold over column.coverage (pileup.d:252). */
int bgp_hist_reduce(bgp_ctx* ctx, void* rccl_comm, uint64_t* device_hist, size_t n);

/* ---- timing hooks for bench.py (HIP events on the ctx's stream) ---------------------------- */
typedef struct bgp_kernel_time { const char* name; double total_ms; uint64_t launches; uint64_t algorithmic_bytes; } bgp_kernel_time;
int bgp_timing_enable(bgp_ctx* ctx, int on);
int bgp_timing_reset(bgp_ctx* ctx);
int bgp_timing_get(bgp_ctx* ctx, const bgp_kernel_time** out, uint32_t* n);

/* ---- synthetic input generator (bench/test support) ------------------------- */
typedef struct bgp_synth_cfg_c {
    uint64_t seed, n_reads;
    uint32_t n_ref, ref_len, len_min, len_max, profile, first_ref;
} bgp_synth_cfg_c;
/* Host generation of records [first, first+count) of a configuration.  out == NULL: size query. */
int bgp_synth_host(const bgp_synth_cfg_c* cfg, uint64_t first, uint64_t count, uint8_t* out, uint64_t cap, uint64_t* bytes);
/* Device generation of the whole configuration straight into HBM; the stream is attached to ctx. */
int bgp_synth_device(bgp_ctx* ctx, const bgp_synth_cfg_c* cfg, uint64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif
