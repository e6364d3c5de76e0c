// common.h -- internals shared by the translation units of libbiodgpu.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/biodgpu.h"

#define BGP_HIP(expr)                                                         \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) { bgp_note_hip_error(e_, #expr, __FILE__, __LINE__); return BGP_EDEVICE; } \
    } while (0)
#define BGP_TRY(expr)              \
    do {                           \
        int rc_ = (expr);          \
        if (rc_ < 0) return rc_;   \
    } while (0)

void bgp_note_hip_error(hipError_t e, const char* what, const char* file, int line);
// No C++ exception crosses the C ABI (a D / C caller cannot catch it: std::terminate would take the host process
// down): every extern "C" entry with a body worth the name runs inside this pair.
int bgp_api_caught(struct bgp_ctx* c, int status, const char* what);
#define BGP_API_TRY try {
#define BGP_API_CATCH(ctx) } catch (const std::bad_alloc&) { return bgp_api_caught(ctx, BGP_ENOMEM, "out of host memory"); } \
                           catch (const std::exception& e_) { return bgp_api_caught(ctx, BGP_EINVAL, e_.what()); } \
                           catch (...) { return bgp_api_caught(ctx, BGP_EINVAL, "unknown C++ exception"); }

// Device memory the library holds through DevBuf, process-wide: bgp_device_memory reports it, and BGP_HBM_CAP (bytes; api.cpp reads the
// environment once) makes an allocation beyond it fail with BGP_ENOMEM -- how the bounded-memory file path (stream.cpp) is held to its word.
struct DevMem { std::atomic<uint64_t> cur{0}, peak{0}, cap{0}; };
DevMem& bgp_devmem();      // api.cpp
// A growable device allocation (never shrinks; reused across bgp_reset).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return BGP_OK;
        DevMem& m = bgp_devmem();
        if (p) { (void)hipFree(p); p = nullptr; m.cur -= cap; cap = 0; }
        size_t want = bytes + (bytes >> 4) + 256;
        const uint64_t lim = m.cap.load();
        if (lim && m.cur.load() + want > lim) { want = bytes + 256; if (m.cur.load() + want > lim) { bgp_note_hip_error(hipErrorOutOfMemory, "BGP_HBM_CAP: the library's device memory would exceed the cap", __FILE__, __LINE__); return BGP_ENOMEM; } }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; bgp_note_hip_error(e, "hipMalloc", __FILE__, __LINE__); return e == hipErrorOutOfMemory ? BGP_ENOMEM : BGP_EDEVICE; }
        cap = want;
        const uint64_t now = (m.cur += want);
        for (uint64_t pk = m.peak.load(); now > pk && !m.peak.compare_exchange_weak(pk, now);) { }
        return BGP_OK;
    }
    void release() { if (p) { (void)hipFree(p); bgp_devmem().cur -= cap; } p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct HostBuf {                      // pinned host memory
    void* p = nullptr;
    size_t cap = 0;
    std::vector<void*> retired;       // buffers a later, larger request replaced: a caller may still hold a view of one (the Python
                                      // binding's copy=False windows, the facades' column.reads), so they live until trim() / release()
    int reserve(size_t bytes) {
        if (bytes <= cap) return BGP_OK;
        void* np = nullptr;
        // geometric: a buffer that replaces one is at least twice its size, so the retired generations together stay below the live
        // one's size -- and they can all be kept until bgp_reset (retire below), as include/biodgpu.h promises
        size_t want = bytes + (bytes >> 3) + 4096;
        if (p && want < 2 * cap) want = 2 * cap;
        hipError_t e = hipHostMalloc(&np, want, hipHostMallocDefault);
        if (e != hipSuccess && want > bytes + (bytes >> 3) + 4096) { (void)hipGetLastError(); want = bytes + (bytes >> 3) + 4096; e = hipHostMalloc(&np, want, hipHostMallocDefault); }      // (pinned memory is short: what is needed, no more)
        // still short: dead generations are given back before BGP_ENOMEM is -- first those older than the last one retired (a view a caller
        // still holds is, by the header's contract, of the buffer this call replaces or of the one before), then that one too
        if (e != hipSuccess && retired.size() > 1) { (void)hipGetLastError(); void* keep = retired.back(); retired.pop_back(); trim(); retired.push_back(keep); e = hipHostMalloc(&np, want, hipHostMallocDefault); }
        if (e != hipSuccess && !retired.empty()) { (void)hipGetLastError(); trim(); e = hipHostMalloc(&np, want, hipHostMallocDefault); }
        if (e != hipSuccess) { bgp_note_hip_error(e, "hipHostMalloc", __FILE__, __LINE__); return BGP_ENOMEM; }
        if (p) retire(p);
        p = np; cap = want;
        return BGP_OK;
    }
    int reserve_keep(size_t bytes, size_t used) {   // grow, preserving the first `used` bytes
        if (bytes <= cap) return BGP_OK;
        void* np = nullptr;
        size_t want = bytes + (bytes >> 1) + 4096;
        hipError_t e = hipHostMalloc(&np, want, hipHostMallocDefault);
        if (e != hipSuccess) { bgp_note_hip_error(e, "hipHostMalloc", __FILE__, __LINE__); return BGP_ENOMEM; }
        if (p) { memcpy(np, p, used); (void)hipHostFree(p); }
        p = np; cap = want;
        return BGP_OK;
    }
    // Every retired generation lives until trim() (bgp_reset) / release(): "a pointer kept too long reads stale bytes, never freed memory"
    // (include/biodgpu.h).  Growth is geometric (reserve above), so they add up to less than the live buffer.
    void retire(void* q) { try { retired.push_back(q); } catch (...) { /* (no room for the note: the buffer stays mapped until the process ends rather than under a caller's view) */ } }
    void trim() { for (void* q : retired) (void)hipHostFree(q); retired.clear(); }       // bgp_reset: every iteration over the old results is over
    void release() { trim(); if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return (T*)p; }
};

// Device status word layout (one uint32 of sticky bits set by kernels).
enum : uint32_t {
    ST_EINVAL = 1u << 0, ST_ETRUNC = 1u << 1, ST_ECIGAR = 1u << 2, ST_EUNSORTED = 1u << 3, ST_ESEQ = 1u << 4, ST_EMD = 1u << 5,
};
int bgp_status_from_bits(uint32_t bits);      // hard errors first; ESEQ is soft
constexpr uint32_t ST_ODDREAD = 1u << 17;      // not an error: some record has more than BGP_PLAN_MAX_OPS CIGAR ops or more than 255 bases -- K4 walks such reads from checkpoints and plans every read itself, so k12_lds writes the first-ops side array, not plan rows
constexpr uint32_t ST_LONGCIG = 1u << 16;      // not an error: some record has more than BGP_LONG_CIGAR ops (its end_position is finished by k1_long_cigar)

struct KernelTimer {
    std::string name;
    double total_ms = 0;
    uint64_t launches = 0;
    uint64_t algorithmic_bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct RefInfo { std::string name; int32_t length; };

struct bgp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int32_t n_ref_hint = 0;

    // staged stream
    HostBuf h_stage; uint64_t h_stage_len = 0;        // pushed host bytes
    struct StageSeg { uint64_t staged; uint32_t len; uint64_t coff; uint32_t skip; };   // staged bytes [staged, +len) = block at coff, from uoffset skip
    std::vector<StageSeg> stage_map;                  // filled by the BGZF loaders (virtual offsets of records)
    bool bai_buildable = false; uint64_t stage_end_coff = 0;   // one whole file staged: createIndex can run (bai_build.hip)
    std::vector<uint8_t> bai_bytes;
    bool region_pending = false; uint32_t region_ref = 0, region_beg = 0, region_end = 0;   // bam[ref][beg..end]: filter applied by bgp_flush
    std::vector<uint64_t> hints;                      // staged offsets where, by the caller's word (first_record_uoffset), a record starts: candidates for the boundary guesser, verified like its own
    DevBuf d_seg_hint;
    DevBuf d_stream_own;                              // device copy of pushed bytes
    hipStream_t io_up = nullptr, io_inf[2] = {nullptr, nullptr};    // the uploads' stream (a file's chunks, a batch's bytes) and the inflate's two, the second also the window prefetch's: made by bgp_init right behind `stream` -- more streams than the device has hardware queues share them, and copies then wait behind a running kernel
    DevBuf d_comp, d_inf_blk; HostBuf h_ring, h_stream_copy;         // device inflate: compressed file, block descriptors + status, pinned read ring
    const uint8_t* d_stream = nullptr; uint64_t stream_len = 0;
    bool stream_on_device = false;                    // d_stream valid
    std::vector<RefInfo> refs;
    DevBuf d_synth;                                   // synthetic stream storage

    // framing (K0)
    uint32_t seg_bytes = 0; uint64_t n_seg = 0;
    DevBuf d_seg_entry, d_seg_end, d_seg_cnt, d_seg_lseq, d_seg_ncig, d_seg_dirty;   // per segment
    DevBuf d_seg_rec_base, d_seg_seq_base, d_seg_cig_base;
    DevBuf d_seg_side; uint32_t side_cap = 0; bool side16 = false;         // per segment: offsets of its records relative to the segment start (written by k0_chase)
    uint32_t seg_bytes_wanted = 0, seg_window_shift = 0, seg_windows = 0;   // K0: segment size asked for (0: its own choice), split into windows of 1 << shift bytes whose prefix counts the chase records (k12_lds); windows per segment of the framing at hand
    DevBuf d_seg_sub;
    bool frame_tolerant_tail = false; uint64_t frame_tail = 0;   // batched staging (batch.cpp): a stream may end inside a record; where the record chain stopped
    bool table_compacted = false;                     // a region filter compacted the record table: records the filter dropped still lie between the kept ones in the stream
    bool defer_long_cigar = false, long_cigar_deferred = false;   // a CIGAR unpack follows the record table at once: it sums the long CIGARs it copies anyway (k2_unpack) instead of k1_long_cigar reading them first
    uint32_t bs0_hint = 0; const uint8_t* bs0_ptr = nullptr; uint64_t bs0_len = 0;   // the first record's block_size as the last framing read it, good while the same bytes are framed again (k0_frame sizes its segments by it: one host round trip less a pass)
    bool odd_reads = false;                           // framing saw a record of more than BGP_PLAN_MAX_OPS CIGAR ops or more than 255 bases (ST_ODDREAD)
    bool table_pending = false, long_cigar = false;   // the record table is written lazily: by k1_table, or by the fused K1+K2 if a decode comes first
    DevBuf d_seg_list; DevBuf d_scan_tmp; DevBuf d_k3_bsum; uint64_t k3_bsum_n = 0; /* K3's scanned tile maxima (k4_bounds mode 3 searches them) and the record count they were made for */ HostBuf h_mail;                // (h_mail: pinned, written by k_mail)
    DevBuf d_sd_desc; int sd_timer_idx = -1;         // single-pass decode (decode.hip: k_stream_decode): control words + one 64-byte descriptor per chunk
    DevBuf d_status;                                  // uint32 status bits + counters
    bool flushed = false;
    uint64_t n_records = 0, sum_lseq = 0, sum_ncig = 0;

    // record table (K1)
    DevBuf d_rec, d_seq_off, d_cigar_off;
    DevBuf d_tilemax; bool tilemax_ok = false;        // per 2048-record tile: maxima of K3's two keys, left by K1 (K3 then skips its own first pass)
    DevBuf d_prow, d_rcls; bool rows_ok = false, lds_table = false;   /* lds_table: the table and all three arenas are k12_lds's, uncompacted (ensure_cig4 can make the side array from them) */      // K4's plan rows (16 B) and classes (1 B) of every record, written by k12_lds (bgp_plan_row below): the accumulate path takes ordinary reads from these, the planner only the rest
    DevBuf d_cig4; bool cig4_ok = false;              // first four raw CIGAR ops of every record (16 B, written by K1): K4's planner reads these, not the stream
    DevBuf d_rec2, d_seq_off2, d_cigar_off2, d_flt[6];   // region filter: compaction targets (swapped in) and scan scratch
    // decode arenas (K2)
    DevBuf d_seq, d_qual, d_cigar;
    uint32_t decoded_what = 0;
    HostBuf h_rec, h_seq_off, h_cigar_off, h_seq, h_qual, h_cigar;

    // pileup (K3/K4)
    DevBuf d_pmax_end, d_pmax_pos;                    // prefix maxima (u64 keys)
    DevBuf d_runs, d_columns, d_depth_hist, d_pstats, d_tiles, d_ck_off, d_ck, d_deep;
    HostBuf h_batch[3]; DevBuf d_batch[2];            // batched staging's pinned and device buffers between two passes (batch.cpp)
    DevBuf d_wlong;                                   // K4: windows touched by a well-formed long read (k4_acc2 walks its ops there)
    DevBuf d_slab, d_wcount, d_wdel, d_wslow;        // K4 plan: per-window descriptor slabs, counts, whole-window '-' levels, slow-read marks
    std::vector<bgp_run> runs;
    uint32_t ck_w = 0;                                // window width the long-read checkpoints (d_ck_off / d_ck) were built for; 0: none
    bgp_pileup_info pinfo{};
    bool piled = false;

    // column iteration
    struct Iter {
        bool active = false; int mode = 0, use_md = 0, skip_zero = 1, want_entries = 0;
        uint64_t start_from = 0, end_at = ~0ull; uint32_t window = 0;
        uint32_t run = 0, run_end = 0; uint64_t next_pos = 0; int32_t ref_id = -1;
        int snp = 0, snp_min_cov = 5, snp_min_wit = 2;
        int maq = 0; float maq_depcorr = 0.17f, maq_eta = 0.03f; uint32_t maq_min_bq = 13;
    } it;
    std::vector<bgp_chunk_info> chunks; int chunks_use_md = 0, chunks_want_entries = 0; uint32_t chunks_window = 0; bool chunks_ready = false;   // pileupChunks: the planned chunks
    HostBuf h_cols, h_entry_off, h_entries, h_refbase, h_snp, h_maq;
    // the window a consumer is likely to ask for next -- the columns behind the one just handed out -- is copied to the host while
    // the consumer works on the current one (api.cpp: emit_window); h_cols2 is the second buffer the two windows alternate between
    struct Prefetch { bool valid = false, compact = false; uint32_t run = 0; uint64_t a = 0, b = 0; hipStream_t stream = nullptr; hipEvent_t done = nullptr; } pf;
    HostBuf h_cols2;
    bool flt_on = false, flt_applied = false; uint32_t flt_require = 0, flt_exclude = 0, flt_min_mapq = 0;      // bgp_set_read_filter: reads the predicate drops cover nothing from the next flush on (region.hip: k_read_filter)
    bool compact_cols = false; DevBuf d_cols8[2]; HostBuf h_cols_wide;      // bgp_set_compact_columns: windows leave as 8 bytes a position (scan.hip: k_pack_columns8) through these staging buffers (the window at hand, the one copied ahead); a window with a count beyond 255 is copied again, wide
    DevBuf d_snp;
    DevBuf d_maq, d_maq_tab; bool maq_tables_ready = false; float maq_tab_depcorr = 0, maq_tab_eta = 0;   // MAQ model (maq.hip): calls of a window; fk | lhet | beta
    DevBuf d_entry_off, d_entries, d_col_lo;
    DevBuf d_md_row, d_md_iv, d_md_state, d_refbase;   // MD path: per-record rows, provider intervals, per-run controller state, dense reference bases
    DevBuf d_md_plong;                                                        // k6_prep: the records whose CIGAR a wave sums (count, then indices)
    DevBuf d_md_long, d_md_ck, d_md_vlist;                                    // long chunks of the provider chain: node list + counters, checkpoints
    DevBuf d_md_succ, d_md_exit, d_md_entry, d_md_bmax;           // parallel provider chain: successor function, per-segment exits, segment entries
    bool md_rows_ready = false, md_rows_full = false;   // K6's rows exist | ... with dna lengths from the MD tags themselves (else: what the CIGARs imply, providers checked later)
    std::vector<uint64_t> md_run_base;                            // per run: index of its start_pos in d_refbase (dense over whole REFERENCES: the provider chain walks or jumps a reference's uncovered stretches, the runs between them are views of it)

    // batched staging (batch.cpp): streams larger than HBM, a batch at a time with the overlapping reads carried over
    struct Batch* batch = nullptr;
    int batch_md = 0, batch_md_skip_zero = 1;          // bgp_batch_set_md: reference_base on every batch's windows, the provider chain carried across the seams
    int batch_maq = 0; float batch_maq_depcorr = 0.f, batch_maq_eta = 0.f; uint32_t batch_maq_min_bq = 0;   // bgp_pileup_maq between bgp_batch_begin and the windows: every batch's windows carry the calls

    // timing
    bool timing = false;
    std::vector<KernelTimer> timers;
    std::vector<bgp_kernel_time> timing_out;

    std::string last_error;
};

// timing helpers: wrap a launch region with HIP events on the ctx stream
struct TimedRegion {
    bgp_ctx* c; int idx = -1; hipEvent_t a = nullptr, b = nullptr; hipStream_t st = nullptr;
    TimedRegion(bgp_ctx* ctx, const char* name, uint64_t alg_bytes, hipStream_t on = nullptr);      // on: the stream the work is put on (default: the ctx's)
    ~TimedRegion();
};
void bgp_timing_collect(bgp_ctx* c);

// ---- per-file entry points (implemented in the .hip units) ----
int k0_frame(bgp_ctx* c);                                   // frame.hip : segments, guess/chase/verify, scans, record table
int k1_table(bgp_ctx* c);                                   // frame.hip : K1 alone (record table + arena offsets)
int k1_finish_long_cigars(bgp_ctx* c);
int k2_decode(bgp_ctx* c, uint32_t what);                   // decode.hip
int k12_fused_decode(bgp_ctx* c, uint32_t what);            // decode.hip : K1 + K2 in one pass over the stream
bool k12_lds_applies(const bgp_ctx* c, uint32_t what); uint32_t k12_lds_segment_bytes(uint64_t stream_len); uint32_t k12_lds_window_shift(); int k12_lds_decode(bgp_ctx* c, uint32_t what);
int ensure_cig4(bgp_ctx* c);                           // the first-ops side array made from the CIGAR arena when k12_lds left it out (decode.hip)   // decode.hip : K1 + K2 through LDS windows, framing from K0
int k012_stream_decode(bgp_ctx* c, uint32_t what);          // decode.hip : K0 + K1 + K2 in ONE pass (framing by look-back); 1 = gave up, take the three-kernel path
constexpr uint32_t BGP_LONG_CIGAR = 32;                      // more ops than this: basesCovered is summed by a whole wave (k1_long_cigar)
int k34_pileup(bgp_ctx* c);                                 // pileup.hip : runs + dense columns
void prefetch_cancel(bgp_ctx* c);      // api.cpp: before d_columns is written again, and when the ctx lets go of its results
int k34_first_survivor(bgp_ctx* c, uint32_t run, uint64_t start_from, uint64_t* pos, uint64_t* record);
int k5_entries(bgp_ctx* c, uint32_t run, uint64_t pos_begin, uint64_t pos_end, uint64_t* n_entries);  // pileup.hip : CSR for a window
int scan_exclusive_u64(bgp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* total_host);   // scan.hip (out has n+1 entries)
int scan_inclusive_max_u64(bgp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n);
int scan_exclusive_u64x3(bgp_ctx* c, const uint64_t* const in[3], uint64_t* const out[3], uint64_t n, uint64_t* totals_host /* [3] or null */);   // three scans, three launches
int pack_columns8(bgp_ctx* c, const bgp_column* in, uint64_t n, uint8_t* out, hipStream_t st);      // scan.hip: n columns as 8 x u8 each, then a u64 that is non-zero when some count was beyond 255
struct FillItem { void* p; uint64_t bytes; uint32_t value; };     // value: the byte every byte becomes
int fill_many(bgp_ctx* c, const FillItem* items, int n_items);   // scan.hip: up to 8 device buffers (16-byte aligned) filled by one launch
struct MailItem { const void* src; void* dst; uint32_t n; };
int mail_fetch(bgp_ctx* c, const MailItem* items, int n_items);   // scan.hip: up to 8 small values from device memory with one kernel and one wait (0 items: just the wait)
int k7_snp_simple(bgp_ctx* c, uint64_t col_index, uint64_t ref_index, uint64_t np, int min_cov, int min_wit);    // md.hip: d_snp[0..np) from d_columns/d_refbase
int k8_maq_calls(bgp_ctx* c, uint64_t col_index, uint64_t ref_index, uint64_t np, int use_md);                    // maq.hip: d_maq[0..np) from the window's entries (K5), the record table and d_refbase
int k6_dna_of_record(bgp_ctx* c, uint64_t i, uint8_t* host_out, uint32_t cap, uint32_t* len, uint32_t* flags);
struct K6Batch {                  // batched staging: the provider chain's state across the seam between two batches (md.hip: k6_chain)
    uint32_t in_have = 0, in_ci = 0, in_y = 0, in_ce = 0; uint64_t in_P = 0, in_i = 0;      // resume the first run from this state (record indices of THIS batch) | chunk_end_position carried in
    uint32_t stop_at = 0;                                                                     // save the last event boundary in front of this position of the LAST run (0: none)
    uint32_t out_have = 0, out_ci = 0, out_y = 0, out_ce = 0; uint64_t out_P = 0, out_i = 0;
    uint32_t err_last_run = 0, err_dep = 0;      // an MD error: in the batch's last run? decided from the reads that start at or before err_dep (final only if the batch holds all of them: err_dep < the cut's position)
};
int k6_reference_bases(bgp_ctx* c, int skip_zero, uint32_t run_first, uint32_t run_end, uint64_t first_record, int* md_err, int64_t* md_err_from, K6Batch* bt = nullptr);   // md.hip: fills d_refbase
int synth_device(bgp_ctx* c, const bgp_synth_cfg_c* cfg, uint64_t* bytes);   // synth.hip
int synth_host(const bgp_synth_cfg_c* cfg, uint64_t first, uint64_t count, uint8_t* out, uint64_t cap, uint64_t* bytes);
int bgzf_load_bam(bgp_ctx* c, const char* path, int n_threads);               // bgzf_host.cpp
int bgp_parse_bam_header(const uint8_t* p, uint64_t avail, std::vector<RefInfo>& refs, uint64_t* records_off, std::string& err);
struct BgzfBlk { uint64_t coff; uint32_t cdata_off, cdata_len, crc, isize; uint64_t uoff; };      // a BGZF block: file offset, where its CDATA lies in it, footer CRC32 / ISIZE, offset of its first byte in the inflated stream
int bgzf_scan_blocks(const uint8_t* f, uint64_t n, uint64_t stop, uint64_t& off, uint64_t& uoff, std::vector<BgzfBlk>& out, std::string& err);   // bgzf_host.cpp: block headers from file offset off while off < stop (inputstream.d:54-199)
int bgzf_inflate_cdata(const uint8_t* cdata, const BgzfBlk& b, uint8_t* dst);      // zlib + size and CRC checks (block.d:127-216)
int bgzf_read_bam_header(int fd, uint64_t fsz, std::vector<RefInfo>& refs, uint64_t* q, std::string& err);
int k_region_filter(bgp_ctx* c, uint32_t ref, uint32_t beg, uint32_t end);    // region.hip
int read_filter_apply(bgp_ctx* c);                                           // region.hip: bgp_set_read_filter's predicate on the record table
int bai_stage_chunks(bgp_ctx* c, const char* path, const bgp_chunk* chunks, uint32_t n, int n_threads);   // bai_host.cpp
int bai_stage_chunks_device(bgp_ctx* c, const char* path, const bgp_chunk* chunks, uint32_t n);      // bai_host.cpp: the same chunk list inflated by k_inflate
int bai_load_header(bgp_ctx* c, const char* path);
int bai_stage_read_at(bgp_ctx* c, const char* path, uint64_t voffset);
int bai_record_voffset(bgp_ctx* c, uint64_t staged_off, uint64_t* voffset);
struct bgp_inflate_block { uint64_t src, dst; uint32_t clen, isize, crc, pad; };   // CDATA at comp + src -> isize bytes at out + dst
int inflate_blocks_device(bgp_ctx* c, const uint8_t* d_comp, uint8_t* d_out, const bgp_inflate_block* d_blk, uint32_t n_blk, int32_t* d_status, uint64_t out_bytes, hipStream_t on = nullptr);   // inflate.hip
int bgzf_load_bam_device(bgp_ctx* c, const char* path);      // bgzf_host.cpp : file -> HBM compressed -> k_inflate
int bgp_default_threads();                                   // bgzf_host.cpp : affinity mask and cgroup quota applied
int bai_build(bgp_ctx* c, std::vector<uint8_t>& out);
int chunks_plan(bgp_ctx* c, uint64_t block_size, uint64_t start_from, uint64_t end_at);
int emit_window(bgp_ctx* c, uint32_t run, uint64_t a, uint64_t b, int want_entries, int use_md, int snp, bgp_columns* o);   // api.cpp: columns [a, b) of a run to the host
int emit_gap_window(bgp_ctx* c, int32_t ref_id, uint64_t a, uint64_t b, int want_entries, int use_md, int snp, int maq, bgp_columns* o);   // api.cpp: an uncovered stretch between two runs of a reference
int batch_cut(bgp_ctx* c, uint64_t* last_key, uint64_t* carry_record, uint64_t* carry_off);   // pileup.hip: last covering read of the staged stream and where its exact halo starts
void batch_release(bgp_ctx* c);   // batch.cpp
void batch_unpark(bgp_ctx* c);    // batch.cpp: a non-batched staging path begins -- the buffers a batched pass parked in the ctx go back to the system
void bgp_staged_copy(uint8_t* dst, const uint8_t* src, uint64_t n);   // batch.cpp: memcpy into pinned staging, large copies over a few threads
int flush_core(bgp_ctx* c, uint32_t what = 0);        // api.cpp: framing + record table (+ the arenas of `what`) of the stream c->d_stream points at   // chunks.hip : fills c->chunks         // bai_build.hip : createIndex over the staged file

// device helpers shared by kernels
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t ld_u32(const uint8_t* p) {   // unaligned little-endian load
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// 16-byte store that does not ask the L2 to keep the line: for outputs written once and read by a later kernel
// (K2's arenas: 12.9 -> 12.3 ms on cfg3, the stream being read no longer competes with them for the cache)
__device__ __forceinline__ void nt_store16(void* p, uint4 v) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
}
// sort key of a reference id: unmapped (-1) sorts last, as in a coordinate-sorted BAM
__device__ __forceinline__ uint64_t ref_key(int32_t ref_id) { return ref_id < 0 ? 0x7FFFFFFFull : (uint64_t)(uint32_t)(ref_id + 1); }
// K3's scan keys of a record: (reference key << 32 | end_position) and (... | position); 0 for reads with basesCovered() == 0
struct K3Key { uint64_t e, p; };
constexpr int K3_TILE_RECORDS = 2048;
__device__ __forceinline__ K3Key k3_key_of(int32_t ref_id, int32_t pos, int32_t end_pos) {
    K3Key k; k.e = 0; k.p = 0;
    if (end_pos != pos) {                                                  // basesCovered() > 0, pileup.d:481,510
        const uint64_t rk = ref_key(ref_id) << 32;
        k.e = rk | (uint32_t)end_pos; k.p = rk | (uint32_t)pos;
    }
    return k;
}
// Is there a plausible record header at stream offset o?  (readrange.d:118-173 reads block_size, then the fixed fields
// of read.d:907-931: the test is what those fields can hold in a well-formed record.)  Only ever a GUESS for where a
// chain of records may be entered; every boundary is verified by the chain itself.
// In two halves, so that a caller can ask for a candidate's second round of loads and for its successor's bytes together (k0_guess):
// block_size and refID (the eight bytes at o, o + 36 <= len) turn nearly every candidate down ...
__device__ __forceinline__ bool plausible_quick(const uint32_t (&h01)[2], uint64_t len, uint64_t o, int32_t ref_bound) {
    const uint32_t bs = h01[0];
    if (bs < 33u || bs > (1u << 30) || (uint64_t)bs > len - o - 4) return false;
    const int32_t ref = (int32_t)h01[1];
    return ref >= -1 && ref < ref_bound;
}
// ... and the rest of the header with the first twelve bytes of the name (h = the 40 bytes at o + 8 when `wide`, o + 48 <= len; else 28 and zeros)
__device__ __forceinline__ bool plausible_rest(const uint8_t* s, uint64_t len, uint64_t o, uint32_t bs, const uint32_t (&h)[10], bool wide, int32_t ref_bound, bool printable_name) {
    const int32_t pos = (int32_t)h[0];
    const uint32_t bmn = h[1], fnc = h[2];
    const int32_t lseq = (int32_t)h[3], nref = (int32_t)h[4], npos = (int32_t)h[5];
    if (nref < -1 || nref >= ref_bound || pos < -1 || npos < -1 || lseq < 0) return false;
    const uint32_t l_name = bmn & 0xFFu, n_cig = fnc & 0xFFFFu;
    if (l_name == 0) return false;
    const uint64_t need = 32ull + l_name + 4ull * n_cig + (uint64_t)((lseq + 1) / 2) + (uint64_t)lseq;
    if (need > bs) return false;
    // ... and printable (SAM spec: [!-?A-~]{1,254}).  This is what tells a header from the same bytes read four or
    // eight bytes early: shifted, block_size lands in the reference field, position in bin_mq_nl, ... all in range,
    // but the "name" then starts in the zero bytes of next_pos / tlen.  (A file whose names are not printable is framed
    // correctly all the same -- its guesses fail, the chain of records is followed serially.)  Eight bytes at a time: a
    // byte b is in [0x21, 0x7E] iff its top bit is clear, b + 0x5F carries into the top bit and b + 0x01 does not.
    const uint32_t nl = l_name - 1;
    auto printable8 = [](unsigned long long v, unsigned long long keep) {
        const unsigned long long H = 0x8080808080808080ull, t7 = v & ~H;
        const unsigned long long bad = (v & H) | (~(t7 + 0x5F5F5F5F5F5F5F5Full) & H) | ((t7 + 0x0101010101010101ull) & H);
        return (bad & keep) == 0;
    };
    if (wide && l_name <= 12u) {                               // the name lies in what has been loaded (bytes 36 .. 47)
        if (((h[7 + (nl >> 2)] >> (8u * (nl & 3u))) & 0xFFu) != 0u) return false;      // read name is NUL terminated
        if (!printable_name) return true;                      // (the second, relaxed pass of the guesser: frame.hip)
        const unsigned long long v0 = ((unsigned long long)h[8] << 32) | h[7];
        if (nl <= 8u) return printable8(v0, nl == 8u ? ~0ull : (1ull << (8u * nl)) - 1ull);
        return printable8(v0, ~0ull) && printable8((unsigned long long)h[9], (1ull << (8u * (nl - 8u))) - 1ull);
    }
    if (s[o + 36 + l_name - 1] != 0) return false;             // read name is NUL terminated
    if (!printable_name) return true;
    const uint8_t* nm = s + o + 36;
    uint32_t i = 0;
    for (; i + 8 <= nl; i += 8) { unsigned long long v; __builtin_memcpy(&v, nm + i, 8); if (!printable8(v, ~0ull)) return false; }
    if (i < nl) {
        if (nl >= 8) { unsigned long long v; __builtin_memcpy(&v, nm + nl - 8, 8); return printable8(v, ~0ull); }           // the last eight (overlapping what was seen)
        if (o + 36 + 8 <= len) { unsigned long long v; __builtin_memcpy(&v, nm, 8); return printable8(v, (1ull << (8 * nl)) - 1); }
        for (; i < nl; ++i) { const uint8_t ch = nm[i]; if (ch < 0x21 || ch > 0x7E) return false; }
    }
    return true;
}
__device__ __forceinline__ void plausible_load_rest(const uint8_t* s, uint64_t len, uint64_t o, uint32_t (&h)[10], bool* wide) {
    *wide = o + 48 <= len;
    if (*wide) __builtin_memcpy(h, s + o + 8, 40);
    else { __builtin_memcpy(h, s + o + 8, 28); h[7] = h[8] = h[9] = 0u; }
}
__device__ __forceinline__ bool plausible(const uint8_t* s, uint64_t len, uint64_t o, int32_t ref_bound, uint32_t* bs_out, bool printable_name = true) {
    if (o + 36 > len) return false;
    // Two rounds of loads instead of one per test: block_size and refID together turn nearly every candidate down (eight bytes a
    // lane); a candidate that passes them gets the rest of its header and the first twelve bytes of its name in one go.  Word by
    // word -- block_size, the fields, the NUL, the name -- the guesser walked eight to ten dependent round trips a segment.
    uint32_t h01[2];
    __builtin_memcpy(h01, s + o, 8);
    *bs_out = h01[0];
    if (!plausible_quick(h01, len, o, ref_bound)) return false;
    uint32_t h[10];                                            // pos, bin_mq_nl, flag_nc, l_seq, next_refID, next_pos, tlen, name[0..11]
    bool wide;
    plausible_load_rest(s, len, o, h, &wide);
    return plausible_rest(s, len, o, h01[0], h, wide, ref_bound, printable_name);
}
// CIGAR_TYPE bit pairs (cigar.d:113-116): bit0 query consuming, bit1 reference consuming
__device__ __forceinline__ uint32_t cig_qc(uint32_t raw) { return (0x3C1A7u >> ((raw & 15u) * 2)) & 1u; }
__device__ __forceinline__ uint32_t cig_rc(uint32_t raw) { return (0x3C1A7u >> ((raw & 15u) * 2 + 1)) & 1u; }
// ---- K4's plan row of a record (written by k12_lds beside the record table, read by k4_acc2; pileup.hip "rows").  An ORDINARY read --
// one or two M / = / X runs with nothing but I / S / D / N / P / H between them, everything within BGP_ROW_REACH positions of its start --
// is 16 bytes:   x  position (pileup.d:377-379: the column it enters)
//                y  [7:0] bases of the first run  [15:8] of the second (0: none)  [24:16] where the second starts, counted from x
//                   [25] the positions between the runs are D / N: '-' (pileup.d:119-121)  [28] the row is there (class 1)
//              z,w  [41:0] nibble address of the first run's first base in the record stream (2 x byte offset + low nibble, read.d:364-383)
//                   [49:42] query bases from there to the second run's first base
// Classes (a byte a record, its low two bits): 0 the read covers nothing (pileup.d:481,510), 1 the row says it all, 2 anything else (the planner's, k4_plan).
// The moulds are k4_plan's `simple` ones, so that a read contributes the same whichever way it takes.
constexpr uint32_t BGP_ROW_REACH = 256;
constexpr uint32_t BGP_PLAN_MAX_OPS = 16;       // the CIGAR ops of a read k4_plan_rest plans (behind plan rows; four at a time); a stream with a longer CIGAR takes no rows (ST_ODDREAD)
constexpr uint32_t BGP_CLS_NO_SKIP = 0x10u;      // beside the class (low two bits): the whole CIGAR was seen (four ops at most) and holds no N -- dna(read) by the CIGAR is end_position - position
__device__ __forceinline__ uint32_t bgp_plan_row(uint32_t pos, uint32_t n_cig, uint32_t l_seq, bool covered, uint4 c4, uint64_t seq_abs, uint64_t stream_len, uint4* row) {
    const uint32_t ops[4] = {c4.x, c4.y, c4.z, c4.w};
    bool seen_ref = false, bad = false, tail_ref = false, gap_del = false;
    uint32_t nseg = 0, p = 0, q = 0, len0 = 0, len1 = 0, r1 = 0, q0 = 0, q1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t op = ops[k], len = op >> 4;
        const bool on = (uint32_t)k < n_cig, rc = on && cig_rc(op), qc = on && cig_qc(op);
        const bool isM = rc && qc, isD = rc && !qc;
        bad |= rc && (len == 0 || (!seen_ref && !qc));
        if (isM) { if (nseg == 0) { len0 = len; q0 = q; } else { len1 = len; r1 = p; q1 = q; } ++nseg; tail_ref = false; }
        if (isD) { tail_ref = true; gap_del = true; }
        seen_ref |= rc;
        p += rc ? len : 0u; q += qc ? len : 0u;
    }
    const bool ok = covered && n_cig >= 1 && n_cig <= 4 && l_seq >= 1 && l_seq <= 255 && seq_abs + (l_seq + 1) / 2 + 16 <= stream_len
                    && !bad && q <= l_seq && nseg >= 1 && nseg <= 2 && !tail_ref && p <= BGP_ROW_REACH;
    if (nseg == 1) r1 = len0;
    const uint64_t na0 = 2ull * seq_abs + q0;
    const uint64_t zw = na0 | ((uint64_t)((q1 - q0) & 0xFFu) << 42);
    *row = ok ? make_uint4(pos, len0 | (len1 << 8) | (r1 << 16) | ((gap_del ? 1u : 0u) << 25) | (1u << 28), (uint32_t)zw, (uint32_t)(zw >> 32)) : make_uint4(0u, 0u, 0u, 0u);
    return ok ? 1u : covered ? 2u : 0u;
}
#endif
