// api.cpp -- the extern "C" surface of libbiodgpu.so (include/biodgpu.h): context, staging,
// result hand-out, column iteration, timing.  All compute is in the .hip units; there is no CPU
// implementation of any kernel in this library.
#include "common.h"

#include <dlfcn.h>
#include <link.h>

#include <algorithm>
#include <cstdio>

static thread_local std::string g_last_hip_error;
DevMem& bgp_devmem() {
    static DevMem m;
    static const bool once = [] { if (const char* e = getenv("BGP_HBM_CAP")) m.cap.store(strtoull(e, nullptr, 0)); return true; }();
    (void)once;
    return m;
}

void bgp_note_hip_error(hipError_t e, const char* what, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_last_hip_error = buf;
    if (getenv("BGP_VERBOSE")) fprintf(stderr, "[biodgpu] %s\n", buf);
}

// Decode implementations that were measured and lost are kept out of the product: they are compiled only with
// `make EXPERIMENTS=1` (-DBGP_EXPERIMENTS), where environment switches select them and tests/test_gpu_single_pass.py runs them
// (LABNOTES.md has their measurements).  The product has ONE path per call: bgp_flush = K0 + k1_rows,
// bgp_decode_* = k2_unpack, bgp_flush_decode(all sections) = K0 + k12_lds (k1_rows + k2_unpack when the records are too long
// for its windows).
#ifdef BGP_EXPERIMENTS
static int fuse_mode() { const char* e = getenv("BGP_FUSE_MODE"); return e ? atoi(e) : 2; }   // 0: chase-based K1 (k1_emit), then K2; 1: K1+K2 fused (k12_fused); 2: k1_rows from the side offsets, then k2_unpack
static int lds_mode() { const char* e = getenv("BGP_LDS_DECODE"); return e ? atoi(e) : 1; }    // 0: bgp_flush_decode uses k1_rows + k2_unpack instead of k12_lds
static int frame_mode() { const char* e = getenv("BGP_FRAME_MODE"); return e ? atoi(e) : 0; }  // 1: framing, table and arenas in one pass (k_stream_decode: exact, 29-31 ms against 17.7)
static int ensure_table(bgp_ctx* c) { return !c->table_pending ? BGP_OK : (fuse_mode() == 0 ? k1_table(c) : k12_fused_decode(c, 0)); }
extern "C" int bgp_experiments_built(void) { return 1; }       // (not part of the ABI: lets the tests see which build they drive)
#else
static constexpr int fuse_mode() { return 2; }
static constexpr int lds_mode() { return 1; }
static constexpr int frame_mode() { return 0; }
static int ensure_table(bgp_ctx* c) { return !c->table_pending ? BGP_OK : k12_fused_decode(c, 0); }
#endif

// framing + record table (+ the arenas of `what`) of the stream c->d_stream points at (bgp_flush / bgp_flush_decode after the upload; a batch of batch.cpp)
int flush_core(bgp_ctx* c, uint32_t what) {
    c->piled = false; c->md_rows_ready = false; c->decoded_what = 0; c->table_compacted = false;
    if (c->region_pending) what = 0;                     // the arenas are made for the records the filter keeps
    const bool decode_only = (what & BGP_DECODE_ONLY) != 0; what &= BGP_FETCH_ALL;
    int rc = 1;
    if (frame_mode() == 1) { rc = k012_stream_decode(c, what); if (rc < 0) return rc; }
    if (rc == 1) {                                       // K0, then K1 and K2
        const bool lds = lds_mode() && (what & BGP_FETCH_ALL) == BGP_FETCH_ALL;
        c->seg_bytes_wanted = lds ? k12_lds_segment_bytes(c->stream_len) : 0; c->seg_window_shift = lds ? k12_lds_window_shift() : 0;
        rc = k0_frame(c);
        c->seg_bytes_wanted = 0; c->seg_window_shift = 0;
        if (rc < 0) return rc;
        c->flushed = true;
        if (lds && k12_lds_applies(c, what)) BGP_TRY(k12_lds_decode(c, what | (decode_only ? BGP_DECODE_ONLY : 0u)));      // rows and arenas from LDS windows: the stream is read once more, not twice
        else {
            c->defer_long_cigar = (what & BGP_FETCH_CIGAR) != 0 && fuse_mode() != 1;      // the unpack below reads every CIGAR anyway
            if (fuse_mode() != 1) BGP_TRY(ensure_table(c));  // K1; only the fused experiment (BGP_FUSE_MODE=1) defers it to the decode
            c->defer_long_cigar = false;
            if (what) { if (c->table_pending) BGP_TRY(k12_fused_decode(c, what)); else BGP_TRY(k2_decode(c, what)); }
            if (c->long_cigar_deferred) { c->long_cigar_deferred = false; BGP_TRY(k1_finish_long_cigars(c)); }      // (not taken up: by its own kernel after all)
        }
    }
    c->flushed = true;
    if (c->region_pending) { BGP_TRY(ensure_table(c)); BGP_TRY(k_region_filter(c, c->region_ref, c->region_beg, c->region_end)); }   // bam[ref][beg..end]
    c->flt_applied = false;
    if (c->flt_on) { BGP_TRY(ensure_table(c)); BGP_TRY(read_filter_apply(c)); }      // reads.filter!(...) in front of the pileup: the reads it drops cover nothing
    return BGP_OK;
}

int bgp_api_caught(bgp_ctx* c, int status, const char* what) {
    try { if (c) c->last_error = what; else g_last_hip_error = what; } catch (...) { }
    return status;
}

int bgp_status_from_bits(uint32_t bits) {
    if (bits & ST_ETRUNC) return BGP_ETRUNC;
    if (bits & ST_EINVAL) return BGP_EINVAL;
    if (bits & ST_EUNSORTED) return BGP_EUNSORTED;
    if (bits & ST_ECIGAR) return BGP_ECIGAR;
    if (bits & ST_EMD) return BGP_EMD;
    if (bits & ST_ESEQ) return BGP_ESEQ;
    return BGP_OK;
}

TimedRegion::TimedRegion(bgp_ctx* ctx, const char* name, uint64_t alg_bytes, hipStream_t on) : c(ctx) {
    st = on ? on : c->stream;
    if (!c->timing) return;
    for (size_t i = 0; i < c->timers.size(); ++i) if (c->timers[i].name == name) idx = (int)i;
    if (idx < 0) { c->timers.emplace_back(); c->timers.back().name = name; idx = (int)c->timers.size() - 1; }
    c->timers[idx].algorithmic_bytes += alg_bytes;
    c->timers[idx].launches += 1;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { idx = -1; return; }
    (void)hipEventRecord(a, st);
}
TimedRegion::~TimedRegion() {
    if (!c->timing || idx < 0) return;
    (void)hipEventRecord(b, st);
    c->timers[idx].pending.emplace_back(a, b);
}
void bgp_timing_collect(bgp_ctx* c) {
    for (auto& t : c->timers) {
        for (auto& ev : t.pending) {
            float ms = 0;
            (void)hipEventSynchronize(ev.second);
            if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) t.total_ms += ms;
            (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
        }
        t.pending.clear();
    }
}

static uint64_t md_ref_index(const bgp_ctx* c, uint32_t run_idx, uint64_t pos) { return (run_idx < c->md_run_base.size() ? c->md_run_base[run_idx] : 0) + (pos - c->runs[run_idx].start_pos); }

void prefetch_cancel(bgp_ctx* c) {
    if (c->pf.stream) (void)hipStreamSynchronize(c->pf.stream);
    c->pf.valid = false;
}

// columns [a, b) of run `run_idx` copied to the host (ctx-owned buffers), with entries / reference bases / SNP verdicts on request
int emit_window(bgp_ctx* c, uint32_t run_idx, uint64_t a, uint64_t b, int want_entries, int use_md, int snp, bgp_columns* o) {
    const bgp_run& r = c->runs[run_idx];
    const uint64_t np = b - a;
    // The columns come from the window copied ahead when the consumer walks the run window by window (every range of the facades does):
    // while it looked at the previous window the copy of this one ran on a stream of its own.  h_cols is always the buffer handed out
    // last -- a caller's view of it stays good until the next call, as the header promises -- and h_cols2 the one being filled.
    static const bool pf_on = !(getenv("BGP_WINDOW_PREFETCH") && atoi(getenv("BGP_WINDOW_PREFETCH")) == 0);
    // Compact transport (bgp_set_compact_columns): a window leaves the device as 8 bytes a position -- packed by k_pack_columns8 into a
    // staging buffer, a flag word behind the last position telling whether some count was beyond 255 -- and only a window that holds
    // such a count is copied again in the 32-byte layout.  The dense array on the device is what it was.
    const bool compact = c->compact_cols;
    auto queue_copy = [&](HostBuf& hb, DevBuf& stage, uint64_t first, uint64_t n, hipStream_t st) -> int {      // columns [first, first + n) of the dense array into hb
        const bgp_column* src = c->d_columns.as<bgp_column>() + first;
        if (!compact) {
            BGP_TRY(hb.reserve(n * sizeof(bgp_column)));
            TimedRegion tw(c, "d2h_window_columns", n * sizeof(bgp_column), st);
            BGP_HIP(hipMemcpyAsync(hb.p, src, n * sizeof(bgp_column), hipMemcpyDeviceToHost, st));
            return BGP_OK;
        }
        BGP_TRY(hb.reserve(n * 8 + 16)); BGP_TRY(stage.reserve(n * 8 + 16));
        BGP_TRY(pack_columns8(c, src, n, stage.as<uint8_t>(), st));
        TimedRegion tw(c, "d2h_window_columns8", n * 8 + 8, st);
        BGP_HIP(hipMemcpyAsync(hb.p, stage.p, n * 8 + 8, hipMemcpyDeviceToHost, st));
        return BGP_OK;
    };
    const bool hit = c->pf.valid && c->pf.run == run_idx && c->pf.a == a && c->pf.b >= b && c->pf.compact == compact;
    if (c->pf.valid && !hit) prefetch_cancel(c);
    const uint64_t pf_len = c->pf.b - c->pf.a;
    if (hit) {
        BGP_HIP(hipEventSynchronize(c->pf.done));
        std::swap(c->h_cols, c->h_cols2);
        c->pf.valid = false;
    } else BGP_TRY(queue_copy(c->h_cols, c->d_cols8[0], r.column_base + (a - r.start_pos), np, c->stream));
    bool wide_again = false;
    if (compact) {                                      // the flag of the window at hand: a count beyond a byte somewhere in it?
        if (!hit) BGP_HIP(hipStreamSynchronize(c->stream));
        // (it lies behind the positions that were PACKED: a window copied ahead may be longer than the one asked for now -- the last window in
        // front of end_at, or of a batch's cut -- and its flag speaks for all of it)
        const uint64_t packed = hit ? pf_len : np;
        uint64_t flag; memcpy(&flag, c->h_cols.as<uint8_t>() + packed * 8, 8);
        wide_again = flag != 0;
    }
    if (pf_on && b < r.end_pos) {                       // the window behind this one, as long as this one
        const uint64_t nb = std::min<uint64_t>(r.end_pos, b + np);
        if (!c->pf.done) BGP_HIP(hipEventCreateWithFlags(&c->pf.done, hipEventDisableTiming));
        BGP_TRY(queue_copy(c->h_cols2, c->d_cols8[1], r.column_base + (b - r.start_pos), nb - b, c->pf.stream));
        BGP_HIP(hipEventRecord(c->pf.done, c->pf.stream));
        c->pf.valid = true; c->pf.run = run_idx; c->pf.a = b; c->pf.b = nb; c->pf.compact = compact;
    }
    if (wide_again) {                                   // (deeper than 255 somewhere: this window in the 32-byte layout after all)
        BGP_TRY(c->h_cols_wide.reserve(np * sizeof(bgp_column)));
        TimedRegion tw(c, "d2h_window_columns", np * sizeof(bgp_column));
        BGP_HIP(hipMemcpyAsync(c->h_cols_wide.p, c->d_columns.as<bgp_column>() + r.column_base + (a - r.start_pos), np * sizeof(bgp_column), hipMemcpyDeviceToHost, c->stream));
    }
    memset(o, 0, sizeof *o);
    const bool maq = c->it.maq != 0;
    if (maq && !want_entries) { uint64_t ne = 0; BGP_TRY(k5_entries(c, run_idx, a, b, &ne)); }     // the MAQ model reads the per-read entries on the device; they are not copied out unless asked for
    if (want_entries) {
        uint64_t ne = 0;
        BGP_TRY(k5_entries(c, run_idx, a, b, &ne));
        BGP_TRY(c->h_entry_off.reserve((np + 1) * 8)); BGP_TRY(c->h_entries.reserve((ne + 1) * sizeof(bgp_entry)));
        BGP_HIP(hipMemcpyAsync(c->h_entry_off.p, c->d_entry_off.p, (np + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        if (ne) BGP_HIP(hipMemcpyAsync(c->h_entries.p, c->d_entries.p, ne * sizeof(bgp_entry), hipMemcpyDeviceToHost, c->stream));
        o->entry_off = c->h_entry_off.as<uint64_t>(); o->entries = c->h_entries.as<bgp_entry>();
    }
    if (use_md) {
        BGP_TRY(c->h_refbase.reserve(np + 1));
        BGP_HIP(hipMemcpyAsync(c->h_refbase.p, c->d_refbase.as<uint8_t>() + md_ref_index(c, run_idx, a), np, hipMemcpyDeviceToHost, c->stream));
        o->reference_base = c->h_refbase.as<uint8_t>();
    }
    if (use_md && snp) {
        BGP_TRY(k7_snp_simple(c, r.column_base + (a - r.start_pos), md_ref_index(c, run_idx, a), np, c->it.snp_min_cov, c->it.snp_min_wit));
        BGP_TRY(c->h_snp.reserve(np + 1));
        BGP_HIP(hipMemcpyAsync(c->h_snp.p, c->d_snp.p, np, hipMemcpyDeviceToHost, c->stream));
        o->snp = c->h_snp.as<uint8_t>();
    }
    if (maq) {
        BGP_TRY(k8_maq_calls(c, r.column_base + (a - r.start_pos), use_md ? md_ref_index(c, run_idx, a) : 0, np, use_md));
        BGP_TRY(c->h_maq.reserve((np + 1) * sizeof(bgp_maq_call)));
        BGP_HIP(hipMemcpyAsync(c->h_maq.p, c->d_maq.p, np * sizeof(bgp_maq_call), hipMemcpyDeviceToHost, c->stream));
        o->maq = c->h_maq.as<bgp_maq_call>();
    }
    BGP_HIP(hipStreamSynchronize(c->stream));
    o->ref_id = r.ref_id; o->n_positions = (uint32_t)np; o->first_position = a;
    if (!compact) o->columns = c->h_cols.as<bgp_column>();
    else if (wide_again) o->columns = c->h_cols_wide.as<bgp_column>();
    else o->columns8 = c->h_cols.as<bgp_column8>();
    return BGP_OK;
}

// positions [a, b) of a reference that no read covers and no run holds: columns of zeros, reference_base 'N' (pileup.d:245-249),
// no entries, isSNP "no" (coverage below any minimum, simple.d:17), no MAQ call (maq.d:447-451)
int emit_gap_window(bgp_ctx* c, int32_t ref_id, uint64_t a, uint64_t b, int want_entries, int use_md, int snp, int maq, bgp_columns* o) {
    const uint64_t np = b - a;
    BGP_HIP(hipStreamSynchronize(c->stream));
    const size_t col_bytes = c->compact_cols ? 8 : sizeof(bgp_column);
    BGP_TRY(c->h_cols.reserve(np * col_bytes + 16));
    memset(c->h_cols.p, 0, np * col_bytes);
    memset(o, 0, sizeof *o);
    if (want_entries) {
        BGP_TRY(c->h_entry_off.reserve((np + 1) * 8)); BGP_TRY(c->h_entries.reserve(sizeof(bgp_entry)));
        memset(c->h_entry_off.p, 0, (np + 1) * 8);
        o->entry_off = c->h_entry_off.as<uint64_t>(); o->entries = c->h_entries.as<bgp_entry>();
    }
    if (use_md) { BGP_TRY(c->h_refbase.reserve(np + 1)); memset(c->h_refbase.p, 'N', np); o->reference_base = c->h_refbase.as<uint8_t>(); }
    if (use_md && snp) { BGP_TRY(c->h_snp.reserve(np + 1)); memset(c->h_snp.p, BGP_SNP_NO, np); o->snp = c->h_snp.as<uint8_t>(); }
    if (maq) {
        BGP_TRY(c->h_maq.reserve((np + 1) * sizeof(bgp_maq_call)));
        bgp_maq_call none{}; none.quality = 0.f; none.genotype = 0xFF; none.reference_base5 = 4; none.flags = 0; none.n_bases = 0;
        for (uint64_t k = 0; k < np; ++k) c->h_maq.as<bgp_maq_call>()[k] = none;
        o->maq = c->h_maq.as<bgp_maq_call>();
    }
    o->ref_id = ref_id; o->n_positions = (uint32_t)np; o->first_position = a;
    if (c->compact_cols) o->columns8 = c->h_cols.as<bgp_column8>(); else o->columns = c->h_cols.as<bgp_column>();
    return BGP_OK;
}

extern "C" {

const char* bgp_strerror(int s) {
    switch (s) {
        case BGP_OK: return "ok";
        case BGP_EOF: return "end of iteration";
        case BGP_EAGAIN: return "more input needed (batched staging)";
        case BGP_EINVAL: return "invalid argument or malformed BAM record";
        case BGP_ETRUNC: return "record runs past the end of the data";
        case BGP_ECIGAR: return "pileup cursor ran off the CIGAR (invalid read)";
        case BGP_EDEVICE: return "HIP device error (no usable gfx950 device?)";
        case BGP_ENOMEM: return "out of memory";
        case BGP_ESTATE: return "call out of order";
        case BGP_EMD: return "MD tag missing or inconsistent / reference_base not available";
        case BGP_EUNSORTED: return "reads are not coordinate sorted";
        case BGP_ESEQ: return "query offset past the end of a read sequence";
        default: return "unknown status";
    }
}

int bgp_abi_version(void) { return BGP_ABI_VERSION; }

int bgp_device_memory(uint64_t* current, uint64_t* peak, int reset_peak) {
    DevMem& m = bgp_devmem();
    if (current) *current = m.cur.load();
    if (peak) *peak = m.peak.load();
    if (reset_peak) m.peak.store(m.cur.load());
    return BGP_OK;
}
int bgp_set_device_memory_cap(uint64_t bytes) { bgp_devmem().cap.store(bytes); return BGP_OK; }
int bgp_set_read_filter(bgp_ctx* c, uint32_t flag_require, uint32_t flag_exclude, uint32_t min_mapping_quality) {
    if (!c) return BGP_EINVAL;
    if (flag_require > 0xFFFFu || flag_exclude > 0xFFFFu || min_mapping_quality > 255u) return BGP_EINVAL;
    c->flt_require = flag_require; c->flt_exclude = flag_exclude; c->flt_min_mapq = min_mapping_quality;
    c->flt_on = flag_require != 0 || flag_exclude != 0 || min_mapping_quality != 0;
    return BGP_OK;
}
int bgp_set_compact_columns(bgp_ctx* c, int on) { if (!c) return BGP_EINVAL; prefetch_cancel(c); c->compact_cols = on != 0; return BGP_OK; }

int bgp_init(int device, bgp_ctx** out) {
    BGP_API_TRY
    if (!out) return BGP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return BGP_EDEVICE;
    BGP_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    BGP_HIP(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950") && !getenv("BGP_ALLOW_ANY_ARCH")) {
        g_last_hip_error = std::string("device is ") + prop.gcnArchName + ", libbiodgpu is built for gfx950 only";
        return BGP_EDEVICE;
    }
    bgp_ctx* c = new (std::nothrow) bgp_ctx();
    if (!c) return BGP_ENOMEM;
    c->device = device;
    // The ctx's FOUR streams, made here and in this order: the device hands its (four) hardware queues to streams as they come, and a
    // fifth stream shares a queue with an earlier one -- a copy then waits behind whatever kernel that one is running.  Made where they
    // were first needed, a bgp_load_bam_device AFTER a batched pass found its uploads sharing a queue with k_inflate: 118 ms a load
    // against 82 (scratch/e2e_gap3.py).  Kernels; every upload (the file's chunks, a batch's bytes); the inflate's two, the second of which
    // also carries the next window's columns to the host while the caller looks at this one (never while a file is loading).
    hipStream_t* four[4] = {&c->stream, &c->io_up, &c->io_inf[0], &c->io_inf[1]};
    for (hipStream_t* s : four)
        if (hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess) {
            *s = nullptr;
            for (hipStream_t* x : four) if (*x) (void)hipStreamDestroy(*x);
            delete c; return BGP_EDEVICE;
        }
    c->own_stream = true;
    c->pf.stream = c->io_inf[1];
    if (c->d_status.reserve(64) < 0) { for (hipStream_t* x : four) (void)hipStreamDestroy(*x); delete c; return BGP_ENOMEM; }
    *out = c;
    return BGP_OK;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

static void release_all(bgp_ctx* c) {
    DevBuf* d[] = {&c->d_stream_own, &c->d_synth, &c->d_seg_entry, &c->d_seg_end, &c->d_seg_cnt, &c->d_seg_lseq, &c->d_seg_ncig, &c->d_seg_dirty,
                   &c->d_seg_rec_base, &c->d_seg_seq_base, &c->d_seg_cig_base, &c->d_seg_side, &c->d_seg_list, &c->d_scan_tmp, &c->d_k3_bsum, &c->d_sd_desc, &c->d_status, &c->d_rec, &c->d_seq_off, &c->d_cigar_off,
                   &c->d_seq, &c->d_qual, &c->d_cigar, &c->d_pmax_end, &c->d_pmax_pos, &c->d_runs, &c->d_columns, &c->d_depth_hist, &c->d_pstats,
                   &c->d_tiles, &c->d_ck_off, &c->d_ck, &c->d_deep, &c->d_wlong, &c->d_entry_off, &c->d_entries, &c->d_col_lo,
                   &c->d_md_row, &c->d_md_iv, &c->d_md_state, &c->d_refbase, &c->d_md_succ, &c->d_md_exit, &c->d_md_entry, &c->d_md_bmax, &c->d_md_plong, &c->d_md_long, &c->d_md_ck, &c->d_md_vlist, &c->d_snp, &c->d_maq, &c->d_maq_tab, &c->d_rec2, &c->d_seq_off2, &c->d_cigar_off2,
                   &c->d_seg_hint, &c->d_seg_sub, &c->d_tilemax, &c->d_cig4, &c->d_prow, &c->d_rcls, &c->d_slab, &c->d_wcount, &c->d_wdel, &c->d_wslow, &c->d_flt[0], &c->d_flt[1], &c->d_flt[2], &c->d_flt[3], &c->d_flt[4], &c->d_flt[5], &c->d_comp, &c->d_inf_blk, &c->d_cols8[0], &c->d_cols8[1], &c->d_batch[0], &c->d_batch[1]};
    for (auto* b : d) b->release();
    HostBuf* h[] = {&c->h_stage, &c->h_rec, &c->h_seq_off, &c->h_cigar_off, &c->h_seq, &c->h_qual, &c->h_cigar, &c->h_cols, &c->h_cols_wide, &c->h_entry_off, &c->h_entries, &c->h_refbase, &c->h_snp, &c->h_maq, &c->h_ring, &c->h_stream_copy, &c->h_mail, &c->h_cols2, &c->h_batch[0], &c->h_batch[1], &c->h_batch[2]};
    for (auto* b : h) b->release();
}

void bgp_destroy(bgp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prefetch_cancel(c);
    if (c->pf.done) (void)hipEventDestroy(c->pf.done);
    // an open batched pass copies on the ctx's upload stream (batch.cpp: Batch::copy_stream = c->io_up) and batch_release waits for it:
    // the pass is closed and the I/O streams drained while they still exist, and only then destroyed
    batch_release(c);
    if (c->io_up) (void)hipStreamSynchronize(c->io_up);
    for (hipStream_t x : c->io_inf) if (x) (void)hipStreamSynchronize(x);
    if (c->io_up) (void)hipStreamDestroy(c->io_up);
    for (hipStream_t x : c->io_inf) if (x) (void)hipStreamDestroy(x);
    bgp_timing_collect(c);
    release_all(c);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int bgp_set_stream(bgp_ctx* c, void* s) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    BGP_HIP(hipStreamSynchronize(c->stream));
    if (s) { if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; } c->stream = (hipStream_t)s; }
    else if (!c->own_stream) { BGP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_set_n_ref(bgp_ctx* c, int32_t n_ref) { if (!c || n_ref < 0) return BGP_EINVAL; c->n_ref_hint = n_ref; return BGP_OK; }

int bgp_reset(bgp_ctx* c) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    BGP_HIP(hipStreamSynchronize(c->stream));
    prefetch_cancel(c);
    batch_release(c);
    for (HostBuf* hb : {&c->h_cols2, &c->h_rec, &c->h_seq_off, &c->h_cigar_off, &c->h_seq, &c->h_qual, &c->h_cigar, &c->h_cols, &c->h_entry_off, &c->h_entries, &c->h_refbase, &c->h_snp, &c->h_maq, &c->h_stream_copy}) hb->trim();
    c->h_stage_len = 0; c->d_stream = nullptr; c->stream_len = 0; c->stream_on_device = false;
    c->flushed = false; c->piled = false; c->md_rows_ready = false; c->table_pending = false; c->n_records = 0; c->decoded_what = 0; c->it = bgp_ctx::Iter();
    c->hints.clear(); c->runs.clear(); c->chunks.clear(); c->chunks_ready = false; c->stage_map.clear(); c->region_pending = false; c->table_compacted = false; c->bai_buildable = false; c->cig4_ok = false; c->rows_ok = false; c->lds_table = false; c->tilemax_ok = false;
    c->refs.clear(); c->n_ref_hint = 0; c->last_error.clear(); c->bai_bytes.clear();      // nothing of the previous file survives: the loaders and bgp_set_n_ref re-establish these
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_push_inflated(bgp_ctx* c, const uint8_t* data, size_t len, uint64_t coffset, uint32_t first_record_uoffset) {
    BGP_API_TRY
    (void)coffset;
    if (!c || (!data && len)) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device) return BGP_ESTATE;
    // boundaries are found and verified on the device (frame.hip); the hint only tells the guesser where to look first
    if (first_record_uoffset != BGP_UOFFSET_UNKNOWN && first_record_uoffset < len) c->hints.push_back(c->h_stage_len + first_record_uoffset);
    c->bai_buildable = false;
    if (!c->h_stage_len) batch_unpark(c);
    BGP_TRY(c->h_stage.reserve_keep(c->h_stage_len + len, c->h_stage_len));
    bgp_staged_copy(c->h_stage.as<uint8_t>() + c->h_stage_len, data, len);
    c->h_stage_len += len;
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_attach_device_stream(bgp_ctx* c, const void* dptr, uint64_t len) {
    BGP_API_TRY
    if (!c || (!dptr && len)) return BGP_EINVAL;
    if (c->flushed || c->h_stage_len) return BGP_ESTATE;
    c->d_stream = (const uint8_t*)dptr; c->stream_len = len; c->stream_on_device = true;
    return BGP_OK;
    BGP_API_CATCH(c)
}

// K1, three ways (measured on cfg3, 100 M reads, K1 + K2 time):
//   0  k1_emit: one lane per segment re-chases the length prefixes and writes its records' rows (9.0 + 12.7 ms)
//   1  k12_fused: rows and unpack in one kernel, the stream is read once (19.2 ms: 90 VGPRs and a three-phase
//      workgroup cost more than the second read of the stream)
//   2  k12_fused<no unpack>: one lane per RECORD from the offsets k0_chase left behind, rows written coalesced,
//      then k2_unpack as before (5.0 + 13.1 ms) -- the default.

static int flush_impl(bgp_ctx* c, uint32_t what) {
    if (!c) return BGP_EINVAL;
    BGP_HIP(hipSetDevice(c->device));
    if (!c->stream_on_device) {
        BGP_TRY(c->d_stream_own.reserve(c->h_stage_len + 64));
        if (c->h_stage_len) BGP_HIP(hipMemcpyAsync(c->d_stream_own.p, c->h_stage.p, c->h_stage_len, hipMemcpyHostToDevice, c->stream));
        c->d_stream = c->d_stream_own.as<uint8_t>(); c->stream_len = c->h_stage_len; c->stream_on_device = true;
    }
    return flush_core(c, what & (BGP_FETCH_ALL | BGP_DECODE_ONLY));
}

int bgp_flush(bgp_ctx* c) {
    BGP_API_TRY
    return flush_impl(c, 0);
    BGP_API_CATCH(c)
}

int bgp_flush_decode(bgp_ctx* c, uint32_t what) {
    BGP_API_TRY
    return flush_impl(c, what);
    BGP_API_CATCH(c)
}

int bgp_num_records(bgp_ctx* c, uint64_t* n) { if (!c || !n) return BGP_EINVAL; if (!c->flushed) return BGP_ESTATE; *n = c->n_records; return BGP_OK; }

int bgp_decode_device(bgp_ctx* c, uint32_t what, bgp_records* o) {
    BGP_API_TRY
    if (!c || !o) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    const uint32_t todo = what & ~c->decoded_what & BGP_FETCH_ALL;
    if (c->table_pending && fuse_mode() == 1) BGP_TRY(k12_fused_decode(c, todo));      // K1 + K2 in one pass
    else { BGP_TRY(ensure_table(c)); if (todo) BGP_TRY(k2_decode(c, todo)); }
    memset(o, 0, sizeof *o);
    o->n = c->n_records; o->rec = c->d_rec.as<bgp_record>(); o->seq_off = c->d_seq_off.as<uint64_t>(); o->cigar_off = c->d_cigar_off.as<uint64_t>();
    o->seq = (what & BGP_FETCH_SEQ) ? c->d_seq.as<uint8_t>() : nullptr;
    o->qual = (what & BGP_FETCH_QUAL) ? c->d_qual.as<uint8_t>() : nullptr;
    o->cigar = (what & BGP_FETCH_CIGAR) ? c->d_cigar.as<uint32_t>() : nullptr;
    o->stream = c->d_stream; o->stream_len = c->stream_len;
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_decode_fetch(bgp_ctx* c, uint32_t what, bgp_records* o) {
    BGP_API_TRY
    bgp_records d;
    BGP_TRY(bgp_decode_device(c, what, &d));
    const uint64_t n = c->n_records;
    BGP_TRY(c->h_rec.reserve((n + 1) * sizeof(bgp_record))); BGP_TRY(c->h_seq_off.reserve((n + 1) * 8)); BGP_TRY(c->h_cigar_off.reserve((n + 1) * 8));
    if (n) BGP_HIP(hipMemcpyAsync(c->h_rec.p, d.rec, n * sizeof(bgp_record), hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipMemcpyAsync(c->h_seq_off.p, d.seq_off, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipMemcpyAsync(c->h_cigar_off.p, d.cigar_off, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    if ((what & BGP_FETCH_SEQ) && c->sum_lseq) { BGP_TRY(c->h_seq.reserve(c->sum_lseq)); BGP_HIP(hipMemcpyAsync(c->h_seq.p, d.seq, c->sum_lseq, hipMemcpyDeviceToHost, c->stream)); }
    if ((what & BGP_FETCH_QUAL) && c->sum_lseq) { BGP_TRY(c->h_qual.reserve(c->sum_lseq)); BGP_HIP(hipMemcpyAsync(c->h_qual.p, d.qual, c->sum_lseq, hipMemcpyDeviceToHost, c->stream)); }
    if ((what & BGP_FETCH_CIGAR) && c->sum_ncig) { BGP_TRY(c->h_cigar.reserve(c->sum_ncig * 4)); BGP_HIP(hipMemcpyAsync(c->h_cigar.p, d.cigar, c->sum_ncig * 4, hipMemcpyDeviceToHost, c->stream)); }
    BGP_HIP(hipStreamSynchronize(c->stream));
    memset(o, 0, sizeof *o);
    o->n = n; o->rec = c->h_rec.as<bgp_record>(); o->seq_off = c->h_seq_off.as<uint64_t>(); o->cigar_off = c->h_cigar_off.as<uint64_t>();
    o->seq = (what & BGP_FETCH_SEQ) ? c->h_seq.as<uint8_t>() : nullptr;
    o->qual = (what & BGP_FETCH_QUAL) ? c->h_qual.as<uint8_t>() : nullptr;
    o->cigar = (what & BGP_FETCH_CIGAR) ? c->h_cigar.as<uint32_t>() : nullptr;
    o->stream = c->h_stage_len ? c->h_stage.as<uint8_t>() : nullptr; o->stream_len = c->h_stage_len;
    if (!c->h_stage_len && (what & BGP_FETCH_STREAM) && c->stream_len) {          // device-inflated or attached stream: copy on request
        BGP_TRY(c->h_stream_copy.reserve(c->stream_len));
        BGP_HIP(hipMemcpyAsync(c->h_stream_copy.p, c->d_stream, c->stream_len, hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        o->stream = c->h_stream_copy.as<uint8_t>(); o->stream_len = c->stream_len;
    }
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_decode_fetch_range(bgp_ctx* c, uint32_t what, uint64_t first, uint64_t count, bgp_records* o) {
    BGP_API_TRY
    bgp_records d;
    BGP_TRY(bgp_decode_device(c, what, &d));
    const uint64_t n = c->n_records;
    if (first > n || count > n - first) return BGP_EINVAL;
    BGP_TRY(c->h_rec.reserve((count + 2) * sizeof(bgp_record))); BGP_TRY(c->h_seq_off.reserve((count + 1) * 8)); BGP_TRY(c->h_cigar_off.reserve((count + 1) * 8));
    // rows first .. first + count (one more than asked for, when there is one: its rec_off ends the stream slice)
    const uint64_t rows = count + (first + count < n ? 1 : 0);
    if (rows) BGP_HIP(hipMemcpyAsync(c->h_rec.p, d.rec + first, rows * sizeof(bgp_record), hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipMemcpyAsync(c->h_seq_off.p, d.seq_off + first, (count + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipMemcpyAsync(c->h_cigar_off.p, d.cigar_off + first, (count + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    const uint64_t* so = c->h_seq_off.as<uint64_t>(); const uint64_t* co = c->h_cigar_off.as<uint64_t>();
    const uint64_t nb = so[count] - so[0], nc = co[count] - co[0];
    if ((what & BGP_FETCH_SEQ) && nb) { BGP_TRY(c->h_seq.reserve(nb)); BGP_HIP(hipMemcpyAsync(c->h_seq.p, d.seq + so[0], nb, hipMemcpyDeviceToHost, c->stream)); }
    if ((what & BGP_FETCH_QUAL) && nb) { BGP_TRY(c->h_qual.reserve(nb)); BGP_HIP(hipMemcpyAsync(c->h_qual.p, d.qual + so[0], nb, hipMemcpyDeviceToHost, c->stream)); }
    if ((what & BGP_FETCH_CIGAR) && nc) { BGP_TRY(c->h_cigar.reserve(nc * 4)); BGP_HIP(hipMemcpyAsync(c->h_cigar.p, d.cigar + co[0], nc * 4, hipMemcpyDeviceToHost, c->stream)); }
    memset(o, 0, sizeof *o);
    if ((what & BGP_FETCH_STREAM) && count) {
        const bgp_record* hr = c->h_rec.as<bgp_record>();
        const uint64_t s0 = hr[0].rec_off, s1 = rows > count ? hr[count].rec_off : c->stream_len;
        if (s1 < s0 || s1 > c->stream_len) return BGP_EINVAL;      // (a compacted table keeps file order: cannot happen on a framed stream)
        BGP_TRY(c->h_stream_copy.reserve(s1 - s0 + 1));
        if (s1 > s0) BGP_HIP(hipMemcpyAsync(c->h_stream_copy.p, c->d_stream + s0, s1 - s0, hipMemcpyDeviceToHost, c->stream));
        o->stream = c->h_stream_copy.as<uint8_t>(); o->stream_len = s1 - s0;
    }
    BGP_HIP(hipStreamSynchronize(c->stream));
    o->n = count; o->rec = c->h_rec.as<bgp_record>(); o->seq_off = so; o->cigar_off = co;
    o->seq = (what & BGP_FETCH_SEQ) ? c->h_seq.as<uint8_t>() : nullptr;
    o->qual = (what & BGP_FETCH_QUAL) ? c->h_qual.as<uint8_t>() : nullptr;
    o->cigar = (what & BGP_FETCH_CIGAR) ? c->h_cigar.as<uint32_t>() : nullptr;
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_pileup_run(bgp_ctx* c, bgp_pileup_info* info) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(ensure_table(c));
    BGP_TRY(k34_pileup(c));
    c->piled = true; c->md_rows_ready = false;
    if (info) *info = c->pinfo;
    return BGP_OK;
    BGP_API_CATCH(c)
}

// bgp_pileup_begin / bgp_pileup_begin_ref: `one_ref` >= 0 restricts makePileup to the reads of that reference
static int pileup_begin_impl(bgp_ctx* c, int mode, int32_t one_ref, int use_md_tag, uint64_t start_from, uint64_t end_at, int skip_zero, int want_entries, uint32_t window) {
    if (!c || (mode != BGP_PILEUP_COLUMNS && mode != BGP_MAKE_PILEUP)) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    if (!c->piled) BGP_TRY(bgp_pileup_run(c, nullptr));
    auto& it = c->it;
    it = bgp_ctx::Iter();
    it.active = true; it.mode = mode; it.use_md = use_md_tag; it.skip_zero = skip_zero; it.want_entries = want_entries;
    it.window = window ? window : 65536u;
    it.run = 0; it.run_end = (uint32_t)c->runs.size(); it.ref_id = -1;
    if (mode == BGP_PILEUP_COLUMNS) { it.start_from = 0; it.end_at = ~0ull; }     // pileupColumns has no clipping (pileup.d:509)
    else {
        it.start_from = start_from; it.end_at = end_at;
        // pileupInstance (pileup.d:482-494): drop leading reads with position + basesCovered < start_from
        // (any reference), then keep only the reference of the first survivor.  Reads are sorted, so the
        // first survivor lives in the first run whose max end_position reaches start_from... unless an
        // earlier read of that run already qualifies, which is the same run.
        uint32_t j = 0;
        if (one_ref >= 0) {                              // the reads of one reference are the whole input
            while (j < it.run_end && (c->runs[j].ref_id != one_ref || c->runs[j].end_pos < start_from)) ++j;
        } else
            while (j < it.run_end && c->runs[j].end_pos < start_from) ++j;
        if (j < it.run_end) {                               // every run of that reference (a reference is several runs where wide stretches of it are uncovered)
            uint32_t e = j + 1;
            while (e < it.run_end && c->runs[e].ref_id == c->runs[j].ref_id) ++e;
            it.run = j; it.run_end = e; it.ref_id = c->runs[j].ref_id;
        } else { it.run = it.run_end; }
    }
    if (it.run < it.run_end) {
        uint64_t first = c->runs[it.run].start_pos, first_record = ~0ull;
        if (mode == BGP_MAKE_PILEUP && it.start_from > first) {
            // the sweep starts at the first surviving read, not at the run's first read (matters for
            // zero-coverage columns when skip_zero_coverage is false, and for the MD provider chain)
            BGP_TRY(k34_first_survivor(c, it.run, it.start_from, &first, &first_record));
            if (first == ~0ull) { it.run = it.run_end; return BGP_OK; }
        }
        it.next_pos = std::max<uint64_t>(first, it.start_from);
        if (use_md_tag) {
            // PileupRangeUsingMdTag (pileup.d:522-654): the whole provider chain is built here, windows slice it.
            int md_err = 0; int64_t md_from = -1;
            BGP_TRY(k6_reference_bases(c, skip_zero, it.run, it.run_end, first_record, &md_err, &md_from));
            if (md_err) {
                // BioD throws from the popFront that trips over the tag; makePileup stops popping at end_at
                // (pileup.d:497-505), so an error further right is never reached.
                const uint64_t reach = std::max(it.start_from, it.end_at);
                if (mode == BGP_PILEUP_COLUMNS || md_from < 0 || (uint64_t)md_from < reach) {
                    it.active = false;
                    c->last_error = "MD tag inconsistent with CIGAR/SEQ while reconstructing reference bases";
                    return md_err;
                }
            }
        }
        if (want_entries && c->pinfo.n_positions) {
            // The window's pinned host buffers are made HERE, by the stream's average depth and a third more (a deeper window still grows
            // them): pinning gigabytes takes a quarter of a second (12 GB/s), and that used to be the first bgp_pileup_next_window's.
            const uint64_t np = std::min<uint64_t>(it.window, c->pinfo.n_positions);
            const double depth = (double)c->pinfo.sum_coverage / (double)c->pinfo.n_positions;
            const uint64_t ne = (uint64_t)((double)np * depth * 1.34) + 4096;
            BGP_TRY(c->h_cols.reserve(np * sizeof(bgp_column)));
            BGP_TRY(c->h_entry_off.reserve((np + 1) * 8));
            if (ne * sizeof(bgp_entry) <= (8ull << 30)) BGP_TRY(c->h_entries.reserve(ne * sizeof(bgp_entry)));
        }
    }
    return BGP_OK;
}

int bgp_pileup_begin(bgp_ctx* c, int mode, int use_md_tag, uint64_t start_from, uint64_t end_at, int skip_zero, int want_entries, uint32_t window) {
    BGP_API_TRY
    return pileup_begin_impl(c, mode, -1, use_md_tag, start_from, end_at, skip_zero, want_entries, window);
    BGP_API_CATCH(c)
}

int bgp_pileup_begin_ref(bgp_ctx* c, int32_t ref_id, int use_md_tag, uint64_t start_from, uint64_t end_at, int skip_zero, int want_entries, uint32_t window) {
    BGP_API_TRY
    if (ref_id < 0) return BGP_EINVAL;
    return pileup_begin_impl(c, BGP_MAKE_PILEUP, ref_id, use_md_tag, start_from, end_at, skip_zero, want_entries, window);
    BGP_API_CATCH(c)
}

int bgp_pileup_chunks_begin(bgp_ctx* c, int use_md_tag, uint64_t block_size, uint64_t start_from, uint64_t end_at, int want_entries, uint32_t window, uint64_t* n_chunks) {
    BGP_API_TRY
    if (!c || !n_chunks) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    c->chunks_ready = false; c->it = bgp_ctx::Iter();
    if (!c->piled) BGP_TRY(bgp_pileup_run(c, nullptr));
    BGP_TRY(chunks_plan(c, block_size, start_from, end_at));
    if (use_md_tag && !c->runs.empty()) {          // the provider chain of the whole stream (pileupColumns' chain); the chunks slice it
        int md_err = 0; int64_t md_from = -1;
        BGP_TRY(k6_reference_bases(c, 1, 0, (uint32_t)c->runs.size(), ~0ull, &md_err, &md_from));
        if (md_err) { c->last_error = "MD tag inconsistent with CIGAR/SEQ while reconstructing reference bases"; return md_err; }
    }
    c->chunks_use_md = use_md_tag; c->chunks_want_entries = want_entries; c->chunks_window = window; c->chunks_ready = true;
    *n_chunks = c->chunks.size();
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_pileup_chunk(bgp_ctx* c, uint64_t k, bgp_chunk_info* out) {
    BGP_API_TRY
    if (!c || !out) return BGP_EINVAL;
    if (!c->chunks_ready || !c->piled) return BGP_ESTATE;
    if (k >= c->chunks.size()) return BGP_EOF;
    const bgp_chunk_info& ci = c->chunks[k];
    *out = ci;
    auto& it = c->it;
    it = bgp_ctx::Iter();
    it.active = true; it.mode = BGP_MAKE_PILEUP; it.use_md = c->chunks_use_md; it.skip_zero = 1; it.want_entries = c->chunks_want_entries;
    it.window = c->chunks_window ? c->chunks_window : 65536u;
    it.start_from = ci.start_position; it.end_at = ci.end_position; it.ref_id = ci.ref_id;
    it.run = it.run_end = (uint32_t)c->runs.size();                             // no columns unless the reference has covering reads
    if (ci.start_position < ci.end_position)
        for (uint32_t j = 0; j < c->runs.size(); ++j)
            if (c->runs[j].ref_id == ci.ref_id) {
                uint32_t e = j + 1;
                while (e < c->runs.size() && c->runs[e].ref_id == ci.ref_id) ++e;
                it.run = j; it.run_end = e; it.next_pos = std::max<uint64_t>(c->runs[j].start_pos, ci.start_position); break;
            }
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_pileup_snp(bgp_ctx* c, int min_cov, int min_wit) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    if (!c->it.active || !c->it.use_md) return BGP_ESTATE;     // findSNPs piles up with reference bases (simple.d:37-41)
    c->it.snp = 1; c->it.snp_min_cov = min_cov; c->it.snp_min_wit = min_wit;
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_pileup_maq(bgp_ctx* c, float depcorr, float eta, uint32_t min_bq) {
    BGP_API_TRY
    if (!c || !(depcorr >= 0.f && depcorr <= 1.f) || !(eta >= 0.f && eta <= 1.f) || min_bq > 255) return BGP_EINVAL;
    if (c->batch) { c->batch_maq = 1; c->batch_maq_depcorr = depcorr; c->batch_maq_eta = eta; c->batch_maq_min_bq = min_bq; return BGP_OK; }   // batched staging: the windows of every batch (reference_base is 'N' there: no use_md_tag across batches)
    if (!c->it.active) return BGP_ESTATE;
    c->it.maq = 1; c->it.maq_depcorr = depcorr; c->it.maq_eta = eta; c->it.maq_min_bq = min_bq;
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_pileup_ref_id(bgp_ctx* c, int32_t* ref_id) { if (!c || !ref_id) return BGP_EINVAL; if (!c->it.active) return BGP_ESTATE; *ref_id = c->it.ref_id; return BGP_OK; }

int bgp_pileup_next_window(bgp_ctx* c, bgp_columns* o) {
    BGP_API_TRY
    if (!c || !o) return BGP_EINVAL;
    auto& it = c->it;
    if (!it.active) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    for (;;) {
        if (it.run >= it.run_end) return BGP_EOF;
        const bgp_run& r = c->runs[it.run];
        const uint64_t hi = std::min<uint64_t>(r.end_pos, it.end_at);
        if (it.next_pos < r.start_pos) {
            // between two runs of one reference lies a stretch without coverage that has no columns on the device.  The sweep walks it
            // column by column when skip_zero_coverage is false (pileup.d:345-397: empty columns, reference_base 'N'): windows of zeros
            const uint64_t a = it.next_pos, b = std::min<uint64_t>(std::min<uint64_t>(r.start_pos, it.end_at), a + it.window);
            if (a >= b) { it.run = it.run_end; return BGP_EOF; }          // end_at lies inside the stretch
            BGP_TRY(emit_gap_window(c, r.ref_id, a, b, it.want_entries, it.use_md, it.use_md && it.snp, it.maq, o));
            it.next_pos = b;
            return BGP_OK;
        }
        if (it.next_pos >= hi) {
            ++it.run;
            if (it.run < it.run_end) {
                const bgp_run& nx = c->runs[it.run];
                const bool walk = !it.skip_zero && nx.ref_id == r.ref_id && it.next_pos >= r.end_pos;      // the same reference goes on behind an uncovered stretch
                it.next_pos = walk ? std::max<uint64_t>(r.end_pos, it.start_from) : std::max<uint64_t>(nx.start_pos, it.start_from);
            }
            continue;
        }
        const uint64_t a = it.next_pos, b = std::min<uint64_t>(hi, a + it.window);
        BGP_TRY(emit_window(c, it.run, a, b, it.want_entries, it.use_md, it.use_md && it.snp, o));
        it.next_pos = b;
        return BGP_OK;
    }
    BGP_API_CATCH(c)
}

int bgp_read_dna(bgp_ctx* c, uint64_t record, char* out, uint32_t cap, uint32_t* len) {
    BGP_API_TRY
    if (!c || !len || (!out && cap)) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    uint32_t flags = 0;
    BGP_TRY(ensure_table(c));
    BGP_TRY(k6_dna_of_record(c, record, (uint8_t*)out, cap, len, &flags));
    if (flags & 4u) { c->last_error = "internal: dna_length disagrees with dna_walk"; return BGP_ESTATE; }
    if (flags & 2u) return BGP_EINVAL;
    return (flags & 1u) ? BGP_EMD : BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_hist_reduce(bgp_ctx* c, void* comm, uint64_t* device_hist, size_t n) {
    BGP_API_TRY
    if (!c || !comm || !device_hist) return BGP_EINVAL;
    // RCCL is resolved at run time so that single-GPU users need no librccl.
    typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
    static allreduce_fn fn = nullptr;
    if (!fn) {
        // The communicator is the caller's: ncclAllReduce has to come from the RCCL that made it.  A librccl the process has mapped already
        // (the caller's -- e.g. torch/lib/librccl.so, whatever its soname) is taken as it is; only a process without one gets the system's.
        struct Found { char path[4096]; int n; } found; found.path[0] = 0; found.n = 0;
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* p) {
            Found* f = (Found*)p;
            const char* nm = info->dlpi_name;
            const char* base = nm ? strrchr(nm, '/') : nullptr; base = base ? base + 1 : nm;
            if (base && strstr(base, "librccl") && strlen(nm) < sizeof f->path) { if (!f->n) strcpy(f->path, nm); ++f->n; }
            return 0; }, &found);
        void* h = found.n ? dlopen(found.path, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD) : nullptr;
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { c->last_error = "librccl not found"; return BGP_EDEVICE; }
        fn = (allreduce_fn)dlsym(h, "ncclAllReduce");
        if (!fn) return BGP_EDEVICE;
    }
    const int ncclUint64 = 5, ncclSum = 0;
    int rc = fn(device_hist, device_hist, n, ncclUint64, ncclSum, comm, c->stream);
    if (rc != 0) { c->last_error = "ncclAllReduce failed"; return BGP_EDEVICE; }
    BGP_HIP(hipStreamSynchronize(c->stream));
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_timing_enable(bgp_ctx* c, int on) { if (!c) return BGP_EINVAL; c->timing = on != 0; return BGP_OK; }
int bgp_timing_reset(bgp_ctx* c) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    BGP_HIP(hipStreamSynchronize(c->stream));
    bgp_timing_collect(c);
    c->timers.clear();
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_timing_get(bgp_ctx* c, const bgp_kernel_time** out, uint32_t* n) {
    BGP_API_TRY
    if (!c || !out || !n) return BGP_EINVAL;
    BGP_HIP(hipStreamSynchronize(c->stream));
    bgp_timing_collect(c);
    c->timing_out.clear();
    for (auto& t : c->timers) c->timing_out.push_back(bgp_kernel_time{t.name.c_str(), t.total_ms, t.launches, t.algorithmic_bytes});
    *out = c->timing_out.data(); *n = (uint32_t)c->timing_out.size();
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_synth_host(const bgp_synth_cfg_c* cfg, uint64_t first, uint64_t count, uint8_t* out, uint64_t cap, uint64_t* bytes) {
    BGP_API_TRY
    if (!cfg || !bytes) return BGP_EINVAL;
    return synth_host(cfg, first, count, out, cap, bytes);
    BGP_API_CATCH((bgp_ctx*)nullptr)
}
int bgp_synth_device(bgp_ctx* c, const bgp_synth_cfg_c* cfg, uint64_t* bytes) {
    BGP_API_TRY
    if (!c || !cfg) return BGP_EINVAL;
    if (c->flushed || c->h_stage_len) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    batch_unpark(c);
    return synth_device(c, cfg, bytes);
    BGP_API_CATCH(c)
}

int bgp_load_bam(bgp_ctx* c, const char* path, int n_threads) {
    BGP_API_TRY
    if (!c || !path) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device) return BGP_ESTATE;
    batch_unpark(c);
    return bgzf_load_bam(c, path, n_threads);
    BGP_API_CATCH(c)
}
int bgp_load_bam_device(bgp_ctx* c, const char* path) {
    BGP_API_TRY
    if (!c || !path) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device || c->h_stage_len) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    batch_unpark(c);
    return bgzf_load_bam_device(c, path);
    BGP_API_CATCH(c)
}
int bgp_load_bam_region(bgp_ctx* c, const char* path, bgp_bai* bai, int32_t ref_id, uint32_t beg, uint32_t end, int n_threads) {
    BGP_API_TRY
    if (!c || !path || !bai) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device || c->h_stage_len) return BGP_ESTATE;
    if (!(beg < end)) { c->last_error = "start must be less than end"; return BGP_EINVAL; }          // reference.d:76
    if (ref_id < 0) { c->last_error = "invalid reference id"; return BGP_EINVAL; }                   // reference.d:78
    BGP_TRY(bai_load_header(c, path));
    const bgp_chunk* ch = nullptr; uint32_t n = 0;
    const int rc = bgp_bai_chunks(bai, ref_id, beg, end, &ch, &n);
    if (rc < 0) { c->last_error = "Invalid reference sequence index"; return rc; }                   // randomaccessmanager.d:205-207
    batch_unpark(c);
    BGP_TRY(bai_stage_chunks(c, path, ch, n, n_threads));
    c->bai_buildable = false;
    c->region_pending = true; c->region_ref = (uint32_t)ref_id; c->region_beg = beg; c->region_end = end;
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_load_bam_region_device(bgp_ctx* c, const char* path, bgp_bai* bai, int32_t ref_id, uint32_t beg, uint32_t end) {
    BGP_API_TRY
    if (!c || !path || !bai) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device || c->h_stage_len) return BGP_ESTATE;
    if (!(beg < end)) { c->last_error = "start must be less than end"; return BGP_EINVAL; }          // reference.d:76
    if (ref_id < 0) { c->last_error = "invalid reference id"; return BGP_EINVAL; }                   // reference.d:78
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(bai_load_header(c, path));
    const bgp_chunk* ch = nullptr; uint32_t n = 0;
    const int rc = bgp_bai_chunks(bai, ref_id, beg, end, &ch, &n);
    if (rc < 0) { c->last_error = "Invalid reference sequence index"; return rc; }                   // randomaccessmanager.d:205-207
    batch_unpark(c);
    BGP_TRY(bai_stage_chunks_device(c, path, ch, n));
    c->bai_buildable = false;
    c->region_pending = true; c->region_ref = (uint32_t)ref_id; c->region_beg = beg; c->region_end = end;
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_filter_region(bgp_ctx* c, int32_t ref_id, uint32_t beg, uint32_t end) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(ensure_table(c));
    c->bai_buildable = false;
    return k_region_filter(c, (uint32_t)ref_id, beg, end);
    BGP_API_CATCH(c)
}
int bgp_load_bam_read_at(bgp_ctx* c, const char* path, uint64_t voffset) {
    BGP_API_TRY
    if (!c || !path) return BGP_EINVAL;
    if (c->flushed || c->stream_on_device) return BGP_ESTATE;
    if (c->refs.empty()) BGP_TRY(bai_load_header(c, path));
    c->bai_buildable = false;
    batch_unpark(c);
    return bai_stage_read_at(c, path, voffset);
    BGP_API_CATCH(c)
}
int bgp_record_voffset(bgp_ctx* c, uint64_t record, uint64_t* voffset) {
    BGP_API_TRY
    if (!c || !voffset) return BGP_EINVAL;
    if (!c->flushed || record >= c->n_records) return c && c->flushed ? BGP_EINVAL : BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(ensure_table(c));
    bgp_record r;
    BGP_HIP(hipMemcpyAsync(&r, c->d_rec.as<bgp_record>() + record, sizeof r, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    return bai_record_voffset(c, r.rec_off, voffset);
    BGP_API_CATCH(c)
}
int bgp_bai_build(bgp_ctx* c, const uint8_t** bytes, uint64_t* n_bytes) {
    BGP_API_TRY
    if (!c || !bytes || !n_bytes) return BGP_EINVAL;
    if (!c->flushed) return BGP_ESTATE;
    if (c->flt_applied) { c->last_error = "createIndex wants every read's end_position: flush without a read filter (bgp_set_read_filter)"; return BGP_ESTATE; }
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(ensure_table(c));
    BGP_TRY(bai_build(c, c->bai_bytes));
    *bytes = c->bai_bytes.data(); *n_bytes = c->bai_bytes.size();
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_n_ref(bgp_ctx* c, int32_t* n) { if (!c || !n) return BGP_EINVAL; *n = (int32_t)c->refs.size(); return BGP_OK; }
int bgp_ref_info(bgp_ctx* c, int32_t id, const char** name, int32_t* length) {
    BGP_API_TRY
    if (!c || id < 0 || (size_t)id >= c->refs.size()) return BGP_EINVAL;
    if (name) *name = c->refs[id].name.c_str();
    if (length) *length = c->refs[id].length;
    return BGP_OK;
    BGP_API_CATCH(c)
}

const char* bgp_last_error(bgp_ctx* c) { return c && !c->last_error.empty() ? c->last_error.c_str() : g_last_hip_error.c_str(); }

}  // extern "C"
