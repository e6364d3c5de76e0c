/**
    bio.std.hts.bam.gpu.pileup -- BioD's pileup interface served by libbiodgpu (MI355X, gfx950).

    Drop this file into BioD as `bio/std/hts/bam/gpu/pileup.d` and link `-L-lbiodgpu` next to `-L-lz` (Makefile:10).
    A caller changes its imports and nothing else (examples/make_pileup.d:1-16):

    ---
    import bio.std.hts.bam.gpu.pileup;              // instead of bio.std.hts.bam.reader + bio.std.hts.bam.pileup
    auto bam = new BamReader("illu_20_chunk.bam");
    auto pileup = makePileup(bam.reads, true);      // line 16 of the example, unchanged
    foreach (column; pileup) { ... column.reference_base, column.coverage, column.bases, column.reads_starting_here ... }
    ---

    Same function names, parameter lists and defaults as bio/std/hts/bam/pileup.d:683-687 (makePileup), :509
    (pileupColumns) and :1011-1012 (pileupChunks); the column type has PileupColumn's properties (pileup.d:236-290) and
    `column.reads[k]` has PileupRead's (pileup.d:86-230) and, as PileupRead does (`alias read this`, pileup.d:89), IS a
    BioD read: an EagerBamRead over the host copy of the record's bytes (read.d:482,1380-1383), so everything a consumer
    asks of it (examples/read_bam_file.d:33-41, snpcallers/simple.d:24-28, snpcallers/maq.d:396-412: name, sequence,
    base_qualities, cigar, position, ref_id, mapping_quality, is_reverse_strand; tags: read["RG"], read.d:1070-1087) is
    answered by the reference's own code.  Every column is computed by the
    HIP kernels behind the C ABI of include/biodgpu.h; this module moves pointers.  The only FFI precedent in BioD is
    zlib (bio/core/utils/zlib.d:6-245): plain pointers and sizes in, int status codes out, turned into exceptions here
    the way block.d:162-185 turns zlib's into ZlibException.

    Storage rule (BioD's own, pileup.d:396,660-661): what `front` hands out is valid until the next `popFront`.

    This image has no D compiler, so this file is kept honest by tests/test_d_binding.py: it parses the `extern (C)`
    block below and include/biodgpu.h and fails when a function's parameter list, a struct's field order, types or
    size, or a constant differ; it also checks that every property the reference's consumers use is defined here.
*/
module bio.std.hts.bam.gpu.pileup;

import bio.std.hts.bam.cigar : CigarOperation;
import bio.std.hts.bam.read : BamRead, EagerBamRead;
import bio.core.base : Base;

import std.algorithm : map;
import std.conv : to;
import std.exception : enforce;
import std.string : fromStringz, toStringz;

// ---------------------------------------------------------------------------------------------------------------------
// include/biodgpu.h, declaration for declaration
// ---------------------------------------------------------------------------------------------------------------------
extern (C) nothrow @nogc {
    enum BGP_ABI_VERSION = 4;

    enum BGP_OK = 0;
    enum BGP_EINVAL = -1;
    enum BGP_ETRUNC = -2;
    enum BGP_ECIGAR = -3;
    enum BGP_EDEVICE = -4;
    enum BGP_ENOMEM = -5;
    enum BGP_ESTATE = -6;
    enum BGP_EMD = -7;
    enum BGP_EUNSORTED = -8;
    enum BGP_ESEQ = -9;
    enum BGP_EOF = 1;
    enum BGP_EAGAIN = 2;

    struct bgp_ctx;

    const(char)* bgp_strerror(int status);
    int bgp_abi_version();
    const(char)* bgp_last_error(bgp_ctx* ctx);

    int bgp_init(int device, bgp_ctx** out_);
    void bgp_destroy(bgp_ctx* ctx);
    int bgp_device_memory(ulong* current, ulong* peak, int reset_peak);
    int bgp_set_device_memory_cap(ulong bytes);
    int bgp_set_stream(bgp_ctx* ctx, void* hip_stream);
    int bgp_set_n_ref(bgp_ctx* ctx, int n_ref);
    int bgp_reset(bgp_ctx* ctx);

    enum BGP_UOFFSET_UNKNOWN = 0xFFFFFFFFu;
    int bgp_push_inflated(bgp_ctx* ctx, const(ubyte)* data, size_t len, ulong coffset, uint first_record_uoffset);
    int bgp_attach_device_stream(bgp_ctx* ctx, const(void)* device_ptr, ulong len);
    int bgp_load_bam(bgp_ctx* ctx, const(char)* path, int n_threads);
    int bgp_load_bam_device(bgp_ctx* ctx, const(char)* path);
    int bgp_n_ref(bgp_ctx* ctx, int* n_ref);
    int bgp_ref_info(bgp_ctx* ctx, int ref_id, const(char)** name, int* length);

    struct bgp_record {
        int ref_id;
        int pos;
        int end_pos;
        uint bin_mq_nl;
        uint flag_nc;
        int l_seq;
        ulong rec_off;
    }

    int bgp_flush(bgp_ctx* ctx);
    int bgp_flush_decode(bgp_ctx* ctx, uint what);
    int bgp_num_records(bgp_ctx* ctx, ulong* n);

    struct bgp_records {
        ulong n;
        const(bgp_record)* rec;
        const(ulong)* seq_off;
        const(ulong)* cigar_off;
        const(ubyte)* seq;
        const(ubyte)* qual;
        const(uint)* cigar;
        const(ubyte)* stream;
        ulong stream_len;
    }

    enum BGP_FETCH_SEQ = 1u;
    enum BGP_FETCH_QUAL = 2u;
    enum BGP_FETCH_CIGAR = 4u;
    enum BGP_FETCH_ALL = 7u;
    enum BGP_FETCH_STREAM = 8u;
    enum BGP_DECODE_ONLY = 16u;
    int bgp_decode_fetch(bgp_ctx* ctx, uint what, bgp_records* out_);
    int bgp_decode_device(bgp_ctx* ctx, uint what, bgp_records* out_device);
    int bgp_decode_fetch_range(bgp_ctx* ctx, uint what, ulong first, ulong count, bgp_records* out_);

    struct bgp_column {
        uint coverage;
        uint n_starting_here;
        uint a;
        uint c;
        uint g;
        uint t;
        uint other;
        uint del;
    }

    struct bgp_entry {
        uint read_ordinal;
        uint cigar_op_offset;
        uint query_offset;
        ushort cigar_op_index;
        ubyte base;
        ubyte qual;
    }

    struct bgp_run {
        int ref_id;
        uint _pad;
        ulong first_record;
        ulong end_record;
        ulong start_pos;
        ulong end_pos;
        ulong column_base;
    }

    struct bgp_pileup_info {
        uint n_runs;
        const(bgp_run)* runs;
        ulong n_positions;
        ulong n_columns;
        ulong sum_coverage;
        uint max_coverage;
        int soft_status;
        ulong n_reads_used;
        const(bgp_column)* device_columns;
        const(ulong)* device_depth_hist;
    }
    enum BGP_DEPTH_BINS = 65536;

    int bgp_pileup_run(bgp_ctx* ctx, bgp_pileup_info* info);

    enum BGP_PILEUP_COLUMNS = 0;
    enum BGP_MAKE_PILEUP = 1;
    struct bgp_columns {
        int ref_id;
        uint n_positions;
        ulong first_position;
        const(bgp_column)* columns;
        const(ulong)* entry_off;
        const(bgp_entry)* entries;
        const(ubyte)* reference_base;
        const(ubyte)* snp;
        const(bgp_maq_call)* maq;
        const(bgp_column8)* columns8;
    }
    struct bgp_column8 {
        ubyte coverage;
        ubyte n_starting_here;
        ubyte a;
        ubyte c;
        ubyte g;
        ubyte t;
        ubyte other;
        ubyte del;
    }
    int bgp_set_compact_columns(bgp_ctx* ctx, int on);
    int bgp_set_read_filter(bgp_ctx* ctx, uint flag_require, uint flag_exclude, uint min_mapping_quality);
    int bgp_pileup_begin(bgp_ctx* ctx, int mode, int use_md_tag, ulong start_from, ulong end_at,
                         int skip_zero_coverage, int want_entries, uint window_positions);
    int bgp_pileup_begin_ref(bgp_ctx* ctx, int ref_id, int use_md_tag, ulong start_from, ulong end_at,
                             int skip_zero_coverage, int want_entries, uint window_positions);
    int bgp_pileup_next_window(bgp_ctx* ctx, bgp_columns* out_);
    int bgp_pileup_ref_id(bgp_ctx* ctx, int* ref_id);

    int bgp_batch_begin(bgp_ctx* ctx, ulong batch_bytes, int want_entries, uint window_positions);
    int bgp_stream_bam_begin(bgp_ctx* ctx, const(char)* path, ulong batch_bytes, int want_entries, uint window_positions);
    int bgp_batch_set_md(bgp_ctx* ctx, int use_md_tag, int skip_zero_coverage);
    int bgp_batch_push(bgp_ctx* ctx, const(ubyte)* data, size_t len, size_t* taken);
    int bgp_batch_push_pinned(bgp_ctx* ctx, const(ubyte)* data, size_t len, size_t* taken);
    int bgp_batch_acquire(bgp_ctx* ctx, ubyte** ptr, size_t* cap);
    int bgp_batch_commit(bgp_ctx* ctx, size_t n);
    int bgp_batch_end(bgp_ctx* ctx);
    int bgp_batch_next_window(bgp_ctx* ctx, bgp_columns* out_);
    int bgp_batch_ordinal_base(bgp_ctx* ctx, ulong* base);

    struct bgp_chunk_info {
        int ref_id;
        uint _pad;
        ulong first_record;
        ulong last_record;
        ulong start_position;
        ulong end_position;
    }
    int bgp_pileup_chunks_begin(bgp_ctx* ctx, int use_md_tag, ulong block_size, ulong start_from, ulong end_at,
                                int want_entries, uint window_positions, ulong* n_chunks);
    int bgp_pileup_chunk(bgp_ctx* ctx, ulong k, bgp_chunk_info* out_);

    enum BGP_SNP_NO = 0;
    enum BGP_SNP_YES = 1;
    enum BGP_SNP_TIE = 2;
    enum BGP_SNP_NEEDS_ENTRIES = 3;
    int bgp_pileup_snp(bgp_ctx* ctx, int minimum_coverage, int minimum_witnesses);

    struct bgp_maq_call {
        float quality;
        ubyte genotype;
        ubyte reference_base5;
        ubyte flags;
        ubyte n_bases;
    }
    enum BGP_MAQ_CALLED = 1u;
    enum BGP_MAQ_VARIANT = 2u;
    enum BGP_MAQ_SUBSAMPLED = 4u;
    int bgp_pileup_maq(bgp_ctx* ctx, float depcorr, float eta, uint minimum_base_quality);
    int bgp_maq_error_model(float depcorr, float eta, double* fk, double* lhet, double* beta);

    int bgp_read_dna(bgp_ctx* ctx, ulong record, char* out_, uint cap, uint* len);

    struct bgp_bai;
    struct bgp_chunk {
        ulong beg;
        ulong end;
    }
    int bgp_bai_open(const(char)* bam_or_bai_path, bgp_bai** out_);
    void bgp_bai_close(bgp_bai* bai);
    int bgp_bai_n_ref(const(bgp_bai)* bai, int* n_ref);
    int bgp_bai_shape(const(bgp_bai)* bai, int ref_id, uint* n_bins, uint* n_chunks, uint* n_intervals);
    int bgp_bai_min_offset(const(bgp_bai)* bai, int ref_id, int position, ulong* voffset);
    int bgp_bai_chunks(bgp_bai* bai, int ref_id, uint beg, uint end, const(bgp_chunk)** chunks, uint* n);
    int bgp_reg2bin(int beg, int end);
    int bgp_bin_can_overlap(uint bin_id, int beg, int end);
    int bgp_bgzf_read_chunks(const(char)* bam_path, const(bgp_chunk)* chunks, uint n, int n_threads,
                             ubyte* out_, ulong cap, ulong* len);
    int bgp_load_bam_region(bgp_ctx* ctx, const(char)* bam_path, bgp_bai* bai, int ref_id, uint beg, uint end, int n_threads);
    int bgp_load_bam_region_device(bgp_ctx* ctx, const(char)* bam_path, bgp_bai* bai, int ref_id, uint beg, uint end);
    int bgp_filter_region(bgp_ctx* ctx, int ref_id, uint beg, uint end);
    int bgp_load_bam_read_at(bgp_ctx* ctx, const(char)* bam_path, ulong voffset);
    int bgp_record_voffset(bgp_ctx* ctx, ulong record, ulong* voffset);

    int bgp_bai_build(bgp_ctx* ctx, const(ubyte)** bytes, ulong* n_bytes);

    int bgp_hist_reduce(bgp_ctx* ctx, void* rccl_comm, ulong* device_hist, size_t n);

    struct bgp_kernel_time {
        const(char)* name;
        double total_ms;
        ulong launches;
        ulong algorithmic_bytes;
    }
    int bgp_timing_enable(bgp_ctx* ctx, int on);
    int bgp_timing_reset(bgp_ctx* ctx);
    int bgp_timing_get(bgp_ctx* ctx, const(bgp_kernel_time)** out_, uint* n);

    struct bgp_synth_cfg_c {
        ulong seed;
        ulong n_reads;
        uint n_ref;
        uint ref_len;
        uint len_min;
        uint len_max;
        uint profile;
        uint first_ref;
    }
    int bgp_synth_host(const(bgp_synth_cfg_c)* cfg, ulong first, ulong count, ubyte* out_, ulong cap, ulong* bytes);
    int bgp_synth_device(bgp_ctx* ctx, const(bgp_synth_cfg_c)* cfg, ulong* bytes);
}

// ---------------------------------------------------------------------------------------------------------------------
// status codes -> exceptions (cf. ZlibException, bio/core/utils/zlib.d:247-273)
// ---------------------------------------------------------------------------------------------------------------------
class BiodGpuException : Exception {
    int code;
    this(int code, bgp_ctx* c, string file = __FILE__, size_t line = __LINE__) {
        super(describe(code, c), file, line);
        this.code = code;
    }
    private static string describe(int code, bgp_ctx* c) {
        string msg = fromStringz(bgp_strerror(code)).idup;
        const(char)* detail = bgp_last_error(c);
        if (detail !is null && detail[0] != 0) msg ~= " [" ~ fromStringz(detail).idup ~ "]";
        return msg;
    }
}

private int chk(int rc, bgp_ctx* c) {
    if (rc < 0) throw new BiodGpuException(rc, c);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// the staged file: one bgp_ctx per (host thread, GPU) -- include/biodgpu.h "Threading"
// ---------------------------------------------------------------------------------------------------------------------
/// Owns a context with one staged, framed record stream (a whole file, or the reads of a region).
final class GpuSource {
    private bgp_ctx* ctx;

    /// `new BamReader(filename)` -> `.reads` (reader.d:127-138, 228-231): the whole file.
    this(string filename, int device = 0) {
        chk(bgp_init(device, &ctx), null);
        scope (failure) { bgp_destroy(ctx); ctx = null; }
        chk(bgp_load_bam(ctx, filename.toStringz, 0), ctx);
        chk(bgp_flush(ctx), ctx);
    }

    /// `bam[ref_id][start .. end]` (reference.d:73-81 -> RandomAccessManager.getReads, randomaccessmanager.d:330-336).
    this(string filename, int ref_id, uint start, uint end, int device = 0) {
        enforce(start < end, "start must be less than end");                   // reference.d:76
        chk(bgp_init(device, &ctx), null);
        scope (failure) { bgp_destroy(ctx); ctx = null; }
        bgp_bai* bai;
        chk(bgp_bai_open(filename.toStringz, &bai), ctx);                      // "BAM index file (.bai) must be provided"
        scope (exit) bgp_bai_close(bai);
        chk(bgp_load_bam_region(ctx, filename.toStringz, bai, ref_id, start, end, 0), ctx);
        chk(bgp_flush(ctx), ctx);                                              // frames, then keeps what BamReadFilter yields
    }

    ~this() {
        if (ctx !is null) bgp_destroy(ctx);
        ctx = null;
    }

    ulong length() @property {
        ulong n;
        chk(bgp_num_records(ctx, &n), ctx);
        return n;
    }

    /// The windows cross the link as 8 bytes a position wherever every count fits a byte; GpuPileup widens them (pileup.d:236-290's numbers either way).
    GpuSource compactTransport(bool on = true) {
        chk(bgp_set_compact_columns(ctx, on ? 1 : 0), ctx);
        return this;
    }

    /// `reads.filter!(r => (r.flag & require) == require && !(r.flag & exclude) && r.mapping_quality >= minMapq)` in front of the pileup
    /// (the flag properties of read.d:137-194): the file is staged again with the predicate applied where the record table is made.
    GpuSource filtered(string filename, uint require, uint exclude, uint minMapq) {
        chk(bgp_set_read_filter(ctx, require, exclude, minMapq), ctx);
        chk(bgp_reset(ctx), ctx);
        chk(bgp_load_bam(ctx, filename.toStringz, 0), ctx);
        chk(bgp_flush(ctx), ctx);
        return this;
    }
}

/// Host copy of the records the reads of the current window belong to (bgp_decode_fetch_range): rows, SEQ / QUAL /
/// CIGAR slices and the records' own bytes (names, tags).  `first` is the stream ordinal of row 0.
private struct GpuRecordWindow {
    bgp_records r;
    ulong first;
}

// ---------------------------------------------------------------------------------------------------------------------
// PileupRead (pileup.d:86-230)
// ---------------------------------------------------------------------------------------------------------------------
/// `read.sequence`: random-access range of Base over the decoded chars (read.d:364-383,438-440; base.d:85).
struct GpuSequence {
    private const(ubyte)[] chars;
    bool empty() const @property { return chars.length == 0; }
    size_t length() const @property { return chars.length; }
    alias opDollar = length;
    Base front() const @property { return Base(cast(char) chars[0]); }
    Base back() const @property { return Base(cast(char) chars[$ - 1]); }
    void popFront() { chars = chars[1 .. $]; }
    void popBack() { chars = chars[0 .. $ - 1]; }
    GpuSequence save() const @property { return GpuSequence(chars); }
    Base opIndex(size_t i) const { return Base(cast(char) chars[i]); }
    GpuSequence opSlice(size_t a, size_t b) const { return GpuSequence(chars[a .. b]); }
    string toString() const { return (cast(const(char)[]) chars).idup; }
}

/// One read of a column.  As BioD's PileupRead (pileup.d:86-89: `Read read; alias read this;`, Read = EagerBamRead) it IS the read:
/// BioD's own BamRead, constructed over the host copy of the record's bytes (read.d:482: `this(ubyte[] chunk)`; the chunk is a slice,
/// `_is_slice`, read.d:876-883 -- a setter duplicates it first, the window's memory is never written), so every BamRead accessor --
/// name, sequence, base_qualities, cigar, flags, strand, cigarString, `read["RG"]` and the other tags (read.d:1070-1087), toSam,
/// to!string -- comes from the reference's own code, not from a re-implementation here.  `end_position` is the device's (K1 computed
/// position + basesCovered() for every record, read.d:1380-1383): the wrapper's field is filled in, its constructor's CIGAR walk is not run.
struct GpuPileupRead {
    private const(bgp_entry)* e;
    private const(GpuRecordWindow)* w;

    ///
    EagerBamRead!BamRead read;
    ///
    alias read this;

    this(const(bgp_entry)* e, const(GpuRecordWindow)* w) {
        this.e = e; this.w = w;
        const(bgp_record)* r = &w.r.rec[idx];
        // the record's body: block_size bytes behind the four of the length prefix (readrange.d:139-163)
        const(ubyte)* p = w.r.stream + cast(size_t)(r.rec_off - w.r.rec[0].rec_off);
        uint block_size = *cast(const(uint)*) p;
        read.read = BamRead(cast(ubyte[]) p[4 .. 4 + block_size]);
        read.end_position = r.end_pos;
    }

    private size_t idx() const { return cast(size_t)(e.read_ordinal - w.first); }
    private ref const(bgp_record) rec() const { return w.r.rec[idx]; }

    /// pileup.d:94
    CigarOperation cigar_operation() const @property { return decoded_cigar[e.cigar_op_index]; }
    /// pileup.d:99
    uint cigar_operation_offset() const @property { return e.cigar_op_offset; }
    /// pileup.d:104-106
    const(CigarOperation)[] cigar_before() const @property { return decoded_cigar[0 .. e.cigar_op_index]; }
    /// pileup.d:109-111
    const(CigarOperation)[] cigar_after() const @property { return decoded_cigar[e.cigar_op_index + 1 .. $]; }
    /// pileup.d:115-122: the base, or '-' when the current operation is D or N
    char current_base() const @property { return cast(char) e.base; }
    /// pileup.d:127-134: the quality, or 255 when the current operation is D or N
    ubyte current_base_quality() const @property { return e.qual; }
    /// pileup.d:146
    int query_offset() const @property { return cast(int) e.query_offset; }

    /// index of the read in the stream (file order)
    ulong read_ordinal() const @property { return e.read_ordinal; }

    // ---- the device's decoded arenas of the same read (what K2 wrote: chars of base.d:85, Phred bytes, raw CIGAR ops), for a
    // consumer that wants them without BamRead's lazy 4-bit range
    /// read.d:225-228: CigarOperation is exactly one uint (cigar.d:58-70)
    const(CigarOperation)[] decoded_cigar() const @property {
        size_t b = cast(size_t)(w.r.cigar_off[idx] - w.r.cigar_off[0]), en = cast(size_t)(w.r.cigar_off[idx + 1] - w.r.cigar_off[0]);
        return cast(const(CigarOperation)[]) w.r.cigar[b .. en];
    }
    /// read.d:438-440
    GpuSequence decoded_sequence() const @property {
        size_t b = cast(size_t)(w.r.seq_off[idx] - w.r.seq_off[0]), en = cast(size_t)(w.r.seq_off[idx + 1] - w.r.seq_off[0]);
        return GpuSequence(w.r.seq[b .. en]);
    }
    /// read.d:468-470
    const(ubyte)[] decoded_base_qualities() const @property {
        size_t b = cast(size_t)(w.r.seq_off[idx] - w.r.seq_off[0]), en = cast(size_t)(w.r.seq_off[idx + 1] - w.r.seq_off[0]);
        return w.r.qual[b .. en];
    }
}

/// `column.reads`: random-access range of the column's reads, in file order (pileup.d:267-269).
struct GpuColumnReads {
    private const(bgp_entry)[] ents;
    private const(GpuRecordWindow)* w;
    bool empty() const @property { return ents.length == 0; }
    size_t length() const @property { return ents.length; }
    alias opDollar = length;
    GpuPileupRead front() const @property { return GpuPileupRead(&ents[0], w); }
    GpuPileupRead back() const @property { return GpuPileupRead(&ents[$ - 1], w); }
    void popFront() { ents = ents[1 .. $]; }
    void popBack() { ents = ents[0 .. $ - 1]; }
    GpuColumnReads save() const @property { return GpuColumnReads(ents, w); }
    GpuPileupRead opIndex(size_t i) const { return GpuPileupRead(&ents[i], w); }
    GpuColumnReads opSlice(size_t a, size_t b) const { return GpuColumnReads(ents[a .. b], w); }
}

// ---------------------------------------------------------------------------------------------------------------------
// PileupColumn (pileup.d:236-290)
// ---------------------------------------------------------------------------------------------------------------------
struct GpuPileupColumn {
    private int _ref_id;
    private ulong _position;
    private const(bgp_column)* c;
    private const(bgp_entry)[] ents;
    private const(GpuRecordWindow)* w;
    private char _refbase = 'N';
    private int _snp = -1;
    private const(bgp_maq_call)* _maq;

    /// pileup.d:252
    size_t coverage() const @property { return c.coverage; }
    /// pileup.d:257
    int ref_id() const @property { return _ref_id; }
    /// pileup.d:262
    ulong position() const @property { return _position; }
    /// pileup.d:245-249: from the MD tags when use_md_tag, else 'N'
    char reference_base() const @property { return _refbase; }
    /// pileup.d:267-269
    GpuColumnReads reads() const @property { return GpuColumnReads(ents, w); }
    /// pileup.d:272-274
    GpuColumnReads reads_starting_here() const @property { return GpuColumnReads(ents[$ - c.n_starting_here .. $], w); }
    /// pileup.d:277-279
    auto bases() const @property { return ents.map!(en => cast(char) en.base); }
    /// pileup.d:282-284
    auto base_qualities() const @property { return ents.map!(en => en.qual); }
    /// pileup.d:287-289
    auto read_qualities() const @property { return reads.map!(r => r.mapping_quality); }

    /// the per-position summary the device keeps (SURVEY 8 a-out): counts of A, C, G, T, any other base, '-'
    uint[6] counts() const @property { return [c.a, c.c, c.g, c.t, c.other, c.del]; }
    /// simple.isSNP(column, settings) (snpcallers/simple.d:15-41) after GpuPileup.callSNPs: BGP_SNP_*; -1 without
    int snp() const @property { return _snp; }
    /// MaqSnpCaller.makeCall(column) (snpcallers/maq.d:443-486) after GpuPileup.callMaq: null without
    const(bgp_maq_call)* maq_call() const @property { return _maq; }
}

// ---------------------------------------------------------------------------------------------------------------------
// the input range of columns (empty / front / popFront, pileup.d:335-397) over device-computed windows
// ---------------------------------------------------------------------------------------------------------------------
final class GpuPileup {
    private GpuSource src;
    private bgp_columns win;
    private bgp_column[] wide;                      // a window that came as columns8, widened
    private GpuRecordWindow recs;
    private uint k;
    private bool done, skipZero, useMd;
    private int mode;
    private long chunk = -1;                       // >= 0: element `chunk` of pileupChunks
    private ulong _start_position, _end_position;
    private int _ref_id = -1;
    private bool wantSnp, wantMaq;
    private int snpCoverage = 5, snpWitnesses = 2;
    private float maqDepcorr = 0.17f, maqEta = 0.03f;
    private uint maqMinBaseQuality = 13;

    /// AbstractPileup.start_position / end_position / ref_id, pileup.d:439-457
    ulong start_position() const @property { return _start_position; }
    ulong end_position() const @property { return _end_position; }
    int ref_id() const @property { return _ref_id; }

    this(GpuSource source, int mode, bool use_md_tag, ulong start_from, ulong end_at, bool skip_zero_coverage) {
        src = source;
        this.mode = mode;
        useMd = use_md_tag;
        _start_position = start_from;
        _end_position = end_at;
        skipZero = skip_zero_coverage;
        restart();
    }

    /// Element `chunk` of pileupChunks: positions the source's window cursor on the chunk (one chunk of a source at a time).
    private this(GpuSource source, ulong chunk) {
        src = source;
        this.chunk = cast(long) chunk;
        mode = BGP_MAKE_PILEUP;
        skipZero = true;
        restart();
    }

    /// findSNPs' settings (simple.d:8-13): every column carries isSNP's verdict (`column.snp`).  Needs use_md_tag;
    /// restarts the iteration.
    GpuPileup callSNPs(int minimum_coverage = 5, int minimum_witnesses = 2) {
        wantSnp = true;
        snpCoverage = minimum_coverage;
        snpWitnesses = minimum_witnesses;
        restart();
        return this;
    }
    /// MaqSnpCaller's settings (maq.d:329-333): every column carries the model's call (`column.maq_call`).  Restarts the iteration.
    GpuPileup callMaq(float depcorr = 0.17f, float eta = 0.03f, uint minimum_base_quality = 13) {
        wantMaq = true;
        maqDepcorr = depcorr;
        maqEta = eta;
        maqMinBaseQuality = minimum_base_quality;
        restart();
        return this;
    }

    private void restart() {
        if (chunk >= 0) {
            bgp_chunk_info ci;
            chk(bgp_pileup_chunk(src.ctx, cast(ulong) chunk, &ci), src.ctx);
            _start_position = ci.start_position;
            _end_position = ci.end_position;
            _ref_id = ci.ref_id;
        } else {
            chk(bgp_pileup_begin(src.ctx, mode, useMd ? 1 : 0, _start_position, _end_position, skipZero ? 1 : 0, 1, 0), src.ctx);
            if (mode == BGP_MAKE_PILEUP) chk(bgp_pileup_ref_id(src.ctx, &_ref_id), src.ctx);
        }
        if (wantSnp) chk(bgp_pileup_snp(src.ctx, snpCoverage, snpWitnesses), src.ctx);
        if (wantMaq) chk(bgp_pileup_maq(src.ctx, maqDepcorr, maqEta, maqMinBaseQuality), src.ctx);
        done = false;
        fetch();
        settle();
    }

    bool empty() @property { return done; }

    GpuPileupColumn front() @property {
        size_t a = cast(size_t) win.entry_off[k], b = cast(size_t) win.entry_off[k + 1];
        return GpuPileupColumn(win.ref_id, win.first_position + k, win.columns + k, win.entries[a .. b], &recs,
                               win.reference_base !is null ? cast(char) win.reference_base[k] : 'N',
                               win.snp !is null ? cast(int) win.snp[k] : -1,
                               win.maq !is null ? win.maq + k : null);
    }

    void popFront() {
        ++k;
        settle();
    }

    private void fetch() {
        done = chk(bgp_pileup_next_window(src.ctx, &win), src.ctx) == BGP_EOF;
        k = 0;
        if (!done && win.columns is null && win.columns8 !is null) {
            // compact transport (bgp_set_compact_columns): a byte a count over the link, widened here; the consumer sees pileup.d:236-290's numbers
            wide.length = win.n_positions;
            foreach (i; 0 .. win.n_positions) {
                auto s = win.columns8[i];
                bgp_column w;
                w.coverage = s.coverage; w.n_starting_here = s.n_starting_here; w.a = s.a; w.c = s.c; w.g = s.g; w.t = s.t; w.other = s.other; w.del = s.del;
                wide[i] = w;
            }
            win.columns = wide.ptr;
        }
        if (done || win.entry_off is null) return;
        size_t ne = cast(size_t) win.entry_off[win.n_positions];
        if (ne == 0) {
            recs = GpuRecordWindow.init;
            return;
        }
        // the reads of the window's columns: a range of the (sorted) stream
        uint lo = uint.max, hi = 0;
        foreach (ref en; win.entries[0 .. ne]) {
            if (en.read_ordinal < lo) lo = en.read_ordinal;
            if (en.read_ordinal > hi) hi = en.read_ordinal;
        }
        chk(bgp_decode_fetch_range(src.ctx, BGP_FETCH_ALL | BGP_FETCH_STREAM, lo, cast(ulong)(hi - lo) + 1, &recs.r), src.ctx);
        recs.first = lo;
    }

    private void settle() {
        // PileupRange.popFront jumps over zero-coverage sites when skip_zero_coverage (pileup.d:388-392)
        while (!done) {
            if (k >= win.n_positions) {
                fetch();
                continue;
            }
            if (skipZero && win.columns[k].coverage == 0) {
                ++k;
                continue;
            }
            return;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// BamReader for callers of this module, and the three pileup functions
// ---------------------------------------------------------------------------------------------------------------------
/// `new BamReader(filename[, taskPool])` (reader.d:127-138): BioD's own reader -- header, reference table, random access,
/// all untouched through `alias this` -- whose `.reads` remembers which file it reads, so that
/// `makePileup(bam.reads, true)` (examples/make_pileup.d:16, verbatim) resolves to the overloads below.
final class BamReader {
    import bio.std.hts.bam.reader : CpuBamReader = BamReader;
    import std.parallelism : TaskPool, taskPool;

    CpuBamReader cpu;
    alias cpu this;

    this(string filename, TaskPool pool = taskPool) { cpu = new CpuBamReader(filename, pool); }

    /// reader.d:228-231.  The CPU range is still there (`alias this`) for every consumer that is not a pileup.
    auto reads() @property {
        auto r = cpu.reads();
        return GpuReads!(typeof(r))(cpu.filename, r);                          // filename: reader.d:171-173
    }

    /// `bam[ref_id][start .. end]` feeding makePileup: the reads overlapping the region, staged through the .bai
    GpuSource region(int ref_id, uint start, uint end) { return new GpuSource(cpu.filename, ref_id, start, end); }
}

/// What `BamReader.reads` of this module returns: BioD's CPU range plus the file it came from.
struct GpuReads(R) {
    string filename;
    R cpuRange;
    alias cpuRange this;
}

/// pileup.d:683-694, same parameter list and defaults.  Any other read range keeps BioD's CPU template
/// (`import bio.std.hts.bam.pileup : makePileup` stays visible for those).
GpuPileup makePileup(R)(GpuReads!R reads, bool use_md_tag = false, ulong start_from = 0, ulong end_at = ulong.max,
                        bool skip_zero_coverage = true) {
    return new GpuPileup(new GpuSource(reads.filename), BGP_MAKE_PILEUP, use_md_tag, start_from, end_at, skip_zero_coverage);
}
/// the same over reads that are already staged (a region: `makePileup(bam.region(id, a, b), false, a, b)`)
GpuPileup makePileup(GpuSource reads, bool use_md_tag = false, ulong start_from = 0, ulong end_at = ulong.max,
                     bool skip_zero_coverage = true) {
    return new GpuPileup(reads, BGP_MAKE_PILEUP, use_md_tag, start_from, end_at, skip_zero_coverage);
}

/// pileup.d:509-519
GpuPileup pileupColumns(R)(GpuReads!R reads, bool use_md_tag = false, bool skip_zero_coverage = true) {
    return new GpuPileup(new GpuSource(reads.filename), BGP_PILEUP_COLUMNS, use_md_tag, 0, ulong.max, skip_zero_coverage);
}
/// ditto
GpuPileup pileupColumns(GpuSource reads, bool use_md_tag = false, bool skip_zero_coverage = true) {
    return new GpuPileup(reads, BGP_PILEUP_COLUMNS, use_md_tag, 0, ulong.max, skip_zero_coverage);
}

/// `pileupColumns(new BamReader(filename).reads, use_md_tag)` with the FILE streamed through the device in bounded memory
/// (bgp_stream_bam_begin: reader threads -> ring of compressed chunks in HBM -> k_inflate -> a batch's kernels -> windows; BioD's own ring of
/// blocks in flight under a pulling consumer, bio/core/bgzf/inputstream.d:386-447,455-483).  An input range of the columns' summaries
/// (coverage, counts, reference_base); `column.reads` is empty -- for the reads of a file that fits the device use `BamReader`.
final class GpuStreamedPileup {
    private bgp_ctx* ctx;
    private bgp_columns win;
    private bgp_column[] wide;
    private GpuRecordWindow recs;
    private uint k;
    private bool done;

    this(string filename, bool use_md_tag = false, ulong batch_bytes = 0, uint window = 0, bool compact = true, int device = 0) {
        chk(bgp_init(device, &ctx), null);
        scope (failure) { bgp_destroy(ctx); ctx = null; }
        chk(bgp_set_compact_columns(ctx, compact ? 1 : 0), ctx);
        chk(bgp_stream_bam_begin(ctx, filename.toStringz, batch_bytes, 0, window), ctx);
        if (use_md_tag) chk(bgp_batch_set_md(ctx, 1, 1), ctx);
        fetch();
        settle();
    }

    ~this() {
        if (ctx !is null) bgp_destroy(ctx);
        ctx = null;
    }

    bool empty() @property { return done; }

    GpuPileupColumn front() @property {
        return GpuPileupColumn(win.ref_id, win.first_position + k, win.columns + k, null, &recs,
                               win.reference_base !is null ? cast(char) win.reference_base[k] : 'N', -1, null);
    }

    void popFront() {
        ++k;
        settle();
    }

    private void fetch() {
        done = chk(bgp_batch_next_window(ctx, &win), ctx) != BGP_OK;
        k = 0;
        if (!done && win.columns is null && win.columns8 !is null) {
            wide.length = win.n_positions;
            foreach (i; 0 .. win.n_positions) {
                auto s = win.columns8[i];
                bgp_column w;
                w.coverage = s.coverage; w.n_starting_here = s.n_starting_here; w.a = s.a; w.c = s.c; w.g = s.g; w.t = s.t; w.other = s.other; w.del = s.del;
                wide[i] = w;
            }
            win.columns = wide.ptr;
        }
    }

    private void settle() {
        while (!done) {
            if (k >= win.n_positions) {
                fetch();
                continue;
            }
            if (win.columns[k].coverage == 0) {
                ++k;
                continue;
            }
            return;
        }
    }
}

/// pileup.d:1011-1015: a random-access range of pileups, one per chunk of `block_size` bytes of reads
/// (chunksConsumingLessThan, splitter.d:65-91; seams as PileupChunkRange sets them, pileup.d:905-913,932-939).  Elements
/// of ONE source are consumed one at a time (a pool of sources gives `taskPool.map` real concurrency).
struct GpuPileupChunks {
    private GpuSource src;
    private ulong lo, hi;
    bool empty() const @property { return lo >= hi; }
    size_t length() const @property { return cast(size_t)(hi - lo); }
    alias opDollar = length;
    GpuPileup front() @property { return new GpuPileup(src, lo); }
    GpuPileup back() @property { return new GpuPileup(src, hi - 1); }
    void popFront() { ++lo; }
    void popBack() { --hi; }
    GpuPileupChunks save() @property { return GpuPileupChunks(src, lo, hi); }
    GpuPileup opIndex(size_t i) { return new GpuPileup(src, lo + i); }
}
/// ditto
GpuPileupChunks pileupChunks(R)(GpuReads!R reads, bool use_md_tag = false, size_t block_size = 16_384_000,
                                ulong start_from = 0, ulong end_at = ulong.max) {
    return pileupChunks(new GpuSource(reads.filename), use_md_tag, block_size, start_from, end_at);
}
/// ditto
GpuPileupChunks pileupChunks(GpuSource reads, bool use_md_tag = false, size_t block_size = 16_384_000,
                             ulong start_from = 0, ulong end_at = ulong.max) {
    ulong n;
    chk(bgp_pileup_chunks_begin(reads.ctx, use_md_tag ? 1 : 0, block_size, start_from, end_at, 1, 0, &n), reads.ctx);
    return GpuPileupChunks(reads, 0, n);
}
