// batch.cpp -- batched staging: a record stream larger than HBM (or simply arriving over time) piled up a batch at a
// time, with the reads that overlap the seam carried over (the staging pipeline of the path).
//
// Reference shape: BgzfInputStream keeps a bounded ring of decompressed blocks in flight (bio/core/bgzf/inputstream.d:
// 386-447) and BamReadRange / PileupRange consume them as a stream; nothing of BioD needs the whole file in memory.
// Here the unit is a BATCH of inflated bytes (any cut points, also inside a record):
//   * two pinned host buffers take the pushes; a full one goes to HBM with an asynchronous copy on its own stream, to
//     a FIXED offset of one of two device buffers -- behind an area reserved for the carry -- so the copy can start
//     while the previous batch is still in its kernels (its carry is not known yet);
//   * a batch = carry + new bytes is framed (the chain of records may stop in front of a partial record: that tail is
//     carried too), decoded and piled up like any stream.  K3's prefix maxima then give, exactly, the first read that
//     still overlaps the position of the batch's LAST covering read (the same halo rule as the windows and the shards):
//     every column before that position is final and handed out; the reads from that first one on -- a contiguous
//     suffix of the batch, so record ordinals stay consecutive -- are copied, device to device, in front of the next
//     batch's bytes;
//   * the last batch hands out everything.
// Columns (and per-read entries, with stream-wide read ordinals) are identical to the one-shot result (tested).
#include "common.h"
#include <cstddef>
#include <thread>
#include <vector>

#include <algorithm>

#include "batch.h"

// The pinned and device buffers a finished batched pass left in the ctx (for the next bgp_batch_begin) are given back as soon as the ctx
// stages a stream any other way: a one-shot load then finds the pinned host memory and the HBM it had before batches were ever used.
void batch_unpark(bgp_ctx* c) {
    if (c->batch) return;
    for (auto& h : c->h_batch) h.release();
    for (auto& d : c->d_batch) d.release();
}

void batch_release(bgp_ctx* c) {
    Batch* b = c->batch;
    if (!b) return;
    if (b->feed) { for (hipStream_t x : c->io_inf) if (x) (void)hipStreamSynchronize(x); feed_release(c, b); }      // (no inflate left reading the ring or writing into a buffer that is about to go; then the readers are stopped and the rings given back)
    if (b->copy_stream) (void)hipStreamSynchronize(b->copy_stream);      // (the ctx's upload stream: not this pass's to destroy)
    // (the pinned buffers and the two device buffers go back to the ctx: the next bgp_batch_begin takes them again -- pinning
    // 2 x 256 MiB anew cost ~70 ms, 40 % of a 10 M-read stream's whole pass)
    for (int q = 0; q < Batch::NH; ++q) {
        if (b->host_free[q]) (void)hipEventDestroy(b->host_free[q]);
        std::swap(b->host[q], c->h_batch[q]); b->host[q].release();
    }
    for (int q = 0; q < Batch::NDV; ++q) {
        if (b->loaded_ev[q]) (void)hipEventDestroy(b->loaded_ev[q]);
        if (q < 2) std::swap(b->dev[q], c->d_batch[q]);
        b->dev[q].release();
    }
    delete b;
    c->batch = nullptr;
    c->frame_tolerant_tail = false; c->batch_maq = 0; c->batch_md = 0; c->batch_md_skip_zero = 1;
}

// the caller's bytes into the pinned buffer: one thread copies ~10 GB/s on the GPU boxes' hosts, which made this memcpy the
// whole of the device-E2E time (the H2D copy, the kernels and the D2H of the columns all hide behind it); large pushes are
// split over a few threads
void bgp_staged_copy(uint8_t* dst, const uint8_t* src, uint64_t n) {
    static const int n_thr = [] { const char* e = getenv("BGP_PUSH_THREADS"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : v > 16 ? 16 : v; }();
    if (n_thr == 1 || n < (16ull << 20)) { memcpy(dst, src, n); return; }
    const uint64_t part = ((n / (uint64_t)n_thr) + 4095) & ~4095ull;
    std::vector<std::thread> th;
    try {
        for (int k = 1; k < n_thr; ++k) {
            const uint64_t o = part * (uint64_t)k;
            if (o >= n) break;
            th.emplace_back([=] { memcpy(dst + o, src + o, std::min<uint64_t>(part, n - o)); });
        }
    } catch (...) {                                                        // no more threads to be had: this one copies what they would have
        const uint64_t done = part * (uint64_t)(th.size() + 1);
        if (done < n) memcpy(dst + done, src + done, n - done);
    }
    memcpy(dst, src, std::min<uint64_t>(part, n));
    for (auto& t : th) t.join();
}

// the oldest full pinned buffer goes to a free device buffer (asynchronous copy); BGP_EAGAIN when both device buffers are taken
static int ship(bgp_ctx* c, Batch* b) {
    if (!b->n_full) return BGP_OK;
    int q = -1;
    for (int k = 0; k < b->n_dev; ++k) if (b->state[k] == Batch::FREE) { q = k; break; }
    if (q < 0) return BGP_EAGAIN;
    const int h = b->fullq[0];
    BGP_TRY(b->dev[q].reserve(b->reserve + b->batch_bytes + 256));
    b->new_off[q] = b->reserve; b->new_len[q] = b->host_len[h];
    if (b->host_len[h]) { TimedRegion th(c, "h2d_batch", b->host_len[h], b->copy_stream);
        BGP_HIP(hipMemcpyAsync(b->dev[q].as<uint8_t>() + b->reserve, b->host[h].p, b->host_len[h], hipMemcpyHostToDevice, b->copy_stream)); }
    BGP_HIP(hipEventRecord(b->loaded_ev[q], b->copy_stream));
    BGP_HIP(hipEventRecord(b->host_free[h], b->copy_stream));
    b->state[q] = Batch::LOADED; b->queue[b->n_queue++] = q;
    b->host_len[h] = 0;
    for (int k = 1; k < b->n_full; ++k) b->fullq[k - 1] = b->fullq[k];
    --b->n_full;
    return BGP_OK;
}
// every full buffer a device buffer is free for
// the caller's page-locked bytes (bgp_batch_push_pinned), a batch_bytes piece into every free device buffer -- behind whatever full pinned
// buffer of the pass's own is still waiting (pushed earlier: piled up earlier)
static int ship_pinned(bgp_ctx* c, Batch* b) {
    while (b->pin_len && !b->n_full) {
        int q = -1;
        for (int k = 0; k < b->n_dev; ++k) if (b->state[k] == Batch::FREE) { q = k; break; }
        if (q < 0) break;
        const uint64_t n = std::min<uint64_t>(b->pin_len, b->batch_bytes);
        BGP_TRY(b->dev[q].reserve(b->reserve + b->batch_bytes + 256));
        b->new_off[q] = b->reserve; b->new_len[q] = n;
        { TimedRegion th(c, "h2d_batch", n, b->copy_stream);
          BGP_HIP(hipMemcpyAsync(b->dev[q].as<uint8_t>() + b->reserve, b->pin_ptr, n, hipMemcpyHostToDevice, b->copy_stream)); }
        BGP_HIP(hipEventRecord(b->loaded_ev[q], b->copy_stream));
        b->state[q] = Batch::LOADED; b->queue[b->n_queue++] = q;
        b->pin_ptr += n; b->pin_len -= n;
    }
    return BGP_OK;
}
static int ship_ready(bgp_ctx* c, Batch* b) {
    while (b->n_full) { const int rc = ship(c, b); if (rc == BGP_EAGAIN) break; if (rc != BGP_OK) return rc; }
    return ship_pinned(c, b);
}
// the buffer that takes the caller's bytes is full (or the stream is over): it joins the queue, and the next one that is not waiting
// there takes its place (none: fill = -1 until a full one has left)
static void next_fill(Batch* b) {
    b->fill = -1;
    for (int k = 0; k < Batch::NH && b->fill < 0; ++k) {
        bool waiting = false;
        for (int j = 0; j < b->n_full; ++j) waiting |= b->fullq[j] == k;
        if (!waiting) b->fill = k;
    }
}
static void close_fill(Batch* b) {
    if (b->fill < 0) return;
    b->fullq[b->n_full++] = b->fill;
    next_fill(b);
}
// full: no room for another inflated BGZF block (64 KiB, constants.d:26-61) -- a caller that writes whole blocks through acquire / commit
// could otherwise be handed a tail it can never use, with no way to move on.  (Batches smaller than that are full when they are full.)
static bool fill_is_full(const Batch* b) {
    if (b->fill < 0 || !b->host_len[b->fill]) return false;
    const uint64_t room = b->batch_bytes - b->host_len[b->fill];
    return room == 0 || (b->batch_bytes > 65536 && room < 65536);
}
// A buffer to write into: a full one is closed and leaves for the device the moment a device buffer is free -- at the commit / push
// that filled it, or when the batch before last gives up its device buffer (process_next).  BGP_EAGAIN: every pinned buffer is full
// and both device buffers are taken -- hand out windows first.  The buffer was the source of a copy a while ago: wait for that copy.
static int fill_ready(bgp_ctx* c, Batch* b, bool whole_blocks = false) {
    // (bgp_batch_acquire never hands out a tail shorter than an inflated BGZF block -- with batches of less than 64 KiB: no tail at all)
    if (fill_is_full(b) || (whole_blocks && b->fill >= 0 && b->host_len[b->fill] && b->batch_bytes - b->host_len[b->fill] < std::min<uint64_t>(b->batch_bytes, 65536))) close_fill(b);
    BGP_TRY(ship_ready(c, b));
    if (b->fill < 0) next_fill(b);
    if (b->fill < 0) return BGP_EAGAIN;
    if (b->host_len[b->fill] == 0) BGP_HIP(hipEventSynchronize(b->host_free[b->fill]));
    return BGP_OK;
}
static int ship_if_full(bgp_ctx* c, Batch* b) {
    if (b->acq < 0 && fill_is_full(b)) close_fill(b);      // (a buffer a caller is still writing into stays where it is)
    return ship_ready(c, b);
}

// the oldest loaded batch becomes the current one: carry in front, framing (tolerant of a partial last record unless it
// is the last batch), record table, pileup, the cut
static int process_next(bgp_ctx* c, Batch* b) {
    const int q = b->queue[0];
    for (int k = 1; k < b->n_queue; ++k) b->queue[k - 1] = b->queue[k];
    --b->n_queue;
    const bool final = b->ended && b->n_queue == 0 && b->n_full == 0 && b->pin_len == 0 && (b->fill < 0 || b->host_len[b->fill] == 0);
    BGP_HIP(hipStreamWaitEvent(c->stream, b->loaded_ev[q], 0));
    if (b->feed && b->new_len[q]) BGP_TRY(feed_check(c, b, q));      // a batch k_inflate made: every block's size and CRC held (block.d:197-207)
    if (b->carry_len > b->new_off[q]) {                              // a carry longer than the room in front of the new bytes: a bigger buffer, once
        DevBuf bigger;
        const uint64_t room = b->carry_len + (b->carry_len >> 2) + 256;
        BGP_TRY(bigger.reserve(room + b->new_len[q] + 256));
        if (b->new_len[q]) BGP_HIP(hipMemcpyAsync(bigger.as<uint8_t>() + room, b->dev[q].as<uint8_t>() + b->new_off[q], b->new_len[q], hipMemcpyDeviceToDevice, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        b->dev[q].release(); b->dev[q] = bigger; b->new_off[q] = room;
        b->reserve = std::max(b->reserve, room);
    }
    uint8_t* base = b->dev[q].as<uint8_t>() + b->new_off[q] - b->carry_len;
    if (b->carry_len) { TimedRegion tc(c, "d2d_carry", b->carry_len);
        BGP_HIP(hipMemcpyAsync(base, c->d_stream + b->carry_off, b->carry_len, hipMemcpyDeviceToDevice, c->stream)); }
    if (b->cur >= 0) { BGP_HIP(hipStreamSynchronize(c->stream)); b->state[b->cur] = Batch::FREE; BGP_TRY(ship_if_full(c, b)); }     // the carry is out: the old batch's buffer can take the next load -- at once, if a full pinned buffer is waiting
    b->cur = q; b->state[q] = Batch::CURRENT; b->cur_final = final; b->any_batch = true;
    c->d_stream = base; c->stream_len = b->carry_len + b->new_len[q]; c->stream_on_device = true;
    c->frame_tolerant_tail = !final;
    c->cig4_ok = false; c->rows_ok = false; c->lds_table = false;
    BGP_TRY(flush_core(c));
    BGP_TRY(k34_pileup(c));
    c->piled = true; c->md_rows_ready = false;
    b->ord_base = b->next_ord_base;
    b->cut = false; b->carry_off = c->frame_tail; b->carry_len = c->stream_len - c->frame_tail;
    b->next_ord_base = b->ord_base + c->n_records;
    if (!final) {
        uint64_t last_key = 0, carry_record = 0, carry_off = 0;
        BGP_TRY(batch_cut(c, &last_key, &carry_record, &carry_off));
        if (last_key) {                                              // the batch holds a covering read: everything before its position is final
            b->cut = true; b->cut_ref = c->runs.back().ref_id; b->cut_pos = (uint32_t)last_key;
            b->carry_off = carry_off; b->carry_len = c->stream_len - carry_off;
            b->next_ord_base = b->ord_base + carry_record;
        }
    } else { b->carry_off = 0; b->carry_len = 0; }
    if (c->batch_md && !c->runs.empty()) {
        // PileupRangeUsingMdTag (pileup.d:522-654) over the batch, its first run resumed where the previous batch's chain stood at its
        // last event in front of the cut, its last run's state saved the same way.  Columns in front of the cut only depend on reads
        // the batch holds, so they are final; the provider and every read the chain has not looked at yet are carried as well.
        K6Batch bt;
        bt.in_ce = b->md.ce;
        if (b->md.have && c->runs[0].ref_id == b->md.ref && b->md.i_ord >= b->ord_base && (b->md.P_ord == ~0ull || b->md.P_ord >= b->ord_base)) {
            bt.in_have = 1; bt.in_P = b->md.P_ord == ~0ull ? ~0ull : b->md.P_ord - b->ord_base; bt.in_i = b->md.i_ord - b->ord_base; bt.in_ci = b->md.ci; bt.in_y = b->md.y;
        }
        bt.stop_at = (!final && b->cut) ? (uint32_t)b->cut_pos : 0u;
        int md_err = 0; int64_t md_from = -1;
        BGP_TRY(k6_reference_bases(c, c->batch_md_skip_zero, 0, (uint32_t)c->runs.size(), ~0ull, &md_err, &md_from, &bt));
        // (an error may be an artefact of the reads the batch does not hold yet -- the next batch looks at those columns again.  What decides
        // is which reads the failed step looked at: a provider fetched at column F is the best of the reads that start at or before F, and
        // of the reads that start AT the cut's position the batch may hold only the first few.  Judged by the column whose popFront raises,
        // F - 1, a fetch AT the cut's position counted as final: found by fuzzing in round 5, one stream in 800)
        (void)md_from;
        if (md_err && (final || !b->cut || !bt.err_last_run || (uint64_t)bt.err_dep < b->cut_pos)) {
            c->last_error = "MD tag inconsistent with CIGAR/SEQ while reconstructing reference bases";
            return md_err;
        }
        b->md.have = false; b->md.ce = bt.out_ce;
        if (!final && b->cut && bt.out_have) {
            b->md.have = true; b->md.ref = c->runs.back().ref_id; b->md.ci = bt.out_ci; b->md.y = bt.out_y;
            b->md.P_ord = bt.out_P == ~0ull ? ~0ull : b->ord_base + bt.out_P; b->md.i_ord = b->ord_base + bt.out_i;
            uint64_t first_needed = bt.out_i;
            if (bt.out_P != ~0ull) first_needed = std::min(first_needed, bt.out_P);
            {   // ... and every read that reaches beyond the saved column: the rows the chain looks at next ask "is the column in front of
                // me covered" (add()'s _prev_coverage, pileup.d:583-588), which the reads that end before the cut still answer
                uint64_t fpos = 0, frec = ~0ull;
                BGP_TRY(k34_first_survivor(c, (uint32_t)c->runs.size() - 1, (uint64_t)bt.out_y + 1, &fpos, &frec));
                if (frec != ~0ull) first_needed = std::min(first_needed, frec);
            }
            const uint64_t carried = b->next_ord_base - b->ord_base;                       // first record the cut carries
            if (first_needed < carried) {                                                   // the chain still needs reads the columns no longer do
                uint64_t off = 0;
                BGP_HIP(hipMemcpyAsync(&off, reinterpret_cast<const uint8_t*>(c->d_rec.as<bgp_record>() + first_needed) + offsetof(bgp_record, rec_off), 8, hipMemcpyDeviceToHost, c->stream));
                BGP_HIP(hipStreamSynchronize(c->stream));
                b->carry_off = off; b->carry_len = c->stream_len - off; b->next_ord_base = b->ord_base + first_needed;
            }
        }
    }
    b->run = 0; b->run_open = false;
    // a file feeds the pass: the next inflate leaves NOW, when this batch's kernels are through -- it then runs beside the copies of this
    // batch's windows and nothing else (launched beside the kernels it starved them of wave slots and LDS: batch.h)
    if (b->feed) BGP_TRY(feed_pump(c, b, false));
    return BGP_OK;
}

extern "C" {

int bgp_batch_begin(bgp_ctx* c, uint64_t batch_bytes, int want_entries, uint32_t window_positions) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    return batch_begin_core(c, batch_bytes, want_entries, window_positions, true);
    BGP_API_CATCH(c)
}

}  // extern "C"

int batch_begin_core(bgp_ctx* c, uint64_t batch_bytes, int want_entries, uint32_t window_positions, bool host_buffers) {
    BGP_HIP(hipSetDevice(c->device));
    const int32_t n_ref_hint = c->n_ref_hint;                                // bgp_set_n_ref before bgp_batch_begin holds for every batch (bgp_reset forgets it)
    BGP_TRY(bgp_reset(c));
    c->n_ref_hint = n_ref_hint;
    Batch* b = new Batch();
    c->batch = b; c->batch_maq = 0;
    b->batch_bytes = std::max<uint64_t>(batch_bytes ? batch_bytes : (256ull << 20), 1ull << 12);
    b->reserve = std::max<uint64_t>(b->batch_bytes / 8, 1ull << 16);         // room for the carry in front of a batch's new bytes (grown when a carry is longer)
    b->want_entries = want_entries; b->window = window_positions ? window_positions : 65536u;
    b->copy_stream = c->io_up;                       // the ctx's upload stream (api.cpp, bgp_init: four streams, a hardware queue each)
    for (int q = 0; q < Batch::NH; ++q) {
        BGP_HIP(hipEventCreateWithFlags(&b->host_free[q], hipEventDisableTiming));
        BGP_HIP(hipEventRecord(b->host_free[q], b->copy_stream));
        std::swap(b->host[q], c->h_batch[q]);                                            // what the last batched pass left (empty the first time)
        if (host_buffers) BGP_TRY(b->host[q].reserve(b->batch_bytes));
    }
    b->n_dev = 2;
    if (!host_buffers) if (const char* e = getenv("BGP_STREAM_BUFFERS")) b->n_dev = std::min<int>(Batch::NDV, std::max(2, atoi(e)));
    for (int q = 0; q < b->n_dev; ++q) { BGP_HIP(hipEventCreateWithFlags(&b->loaded_ev[q], hipEventDisableTiming)); if (q < 2) std::swap(b->dev[q], c->d_batch[q]); }
    return BGP_OK;
}

extern "C" {

int bgp_batch_push(bgp_ctx* c, const uint8_t* data, size_t len, size_t* taken) {
    BGP_API_TRY
    if (!c || (!data && len) || !taken) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b || b->ended || b->feed) return BGP_ESTATE;      // (a pass bgp_stream_bam_begin feeds from a file takes no bytes from the caller)
    if (b->acq >= 0) return BGP_ESTATE;                  // an acquired buffer is waiting for its commit: a push could ship it from under the caller
    BGP_HIP(hipSetDevice(c->device));
    *taken = 0;
    if (b->pin_len) { BGP_TRY(ship_ready(c, b)); if (b->pin_len) return BGP_EAGAIN; }      // (page-locked bytes pushed earlier go first)
    while (*taken < len) {
        if (b->fill >= 0 && b->host_len[b->fill] == b->batch_bytes) close_fill(b);
        { const int rc = fill_ready(c, b); if (rc != BGP_OK) return rc; }      // BGP_EAGAIN: hand out windows first
        const int h = b->fill;
        const uint64_t n = std::min<uint64_t>(len - *taken, b->batch_bytes - b->host_len[h]);
        bgp_staged_copy(b->host[h].as<uint8_t>() + b->host_len[h], data + *taken, n);
        b->host_len[h] += n; *taken += n;
    }
    if (b->fill >= 0 && b->host_len[b->fill] == b->batch_bytes) { close_fill(b); BGP_TRY(ship_ready(c, b)); }
    return BGP_OK;
    BGP_API_CATCH(c)
}

/* bgp_batch_push for bytes that already lie in page-locked host memory (hipHostMalloc / hipHostRegister): they go to the device from
 * where they are, a batch_bytes piece a device buffer -- no copy into the pass's own pinned buffers.  SURVEY 8(d) (ii): "device E2E
 * from pinned host inflated buffers".  BGP_EAGAIN with *taken < len: both device buffers are taken -- take a window, push the rest. */
int bgp_batch_push_pinned(bgp_ctx* c, const uint8_t* data, size_t len, size_t* taken) {
    BGP_API_TRY
    if (!c || (!data && len) || !taken) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b || b->ended || b->feed) return BGP_ESTATE;
    if (b->acq >= 0) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    *taken = 0;
    if (!len) return BGP_OK;
    {   // page-locked, or refused: a pageable source would make every copy a synchronous, staged one behind the caller's back
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, data) != hipSuccess || at.type != hipMemoryTypeHost) { (void)hipGetLastError(); c->last_error = "bgp_batch_push_pinned: the bytes are not in page-locked host memory (hipHostMalloc / hipHostRegister)"; return BGP_EINVAL; }
    }
    // what was pushed before comes first: a partly filled buffer of the pass's own is closed and leaves in front of these bytes; bytes of an
    // earlier bgp_batch_push_pinned still waiting are followed only by bytes that lie right behind them
    if (b->pin_len) {
        BGP_TRY(ship_ready(c, b));
        if (b->pin_len && b->pin_ptr + b->pin_len != data) return BGP_EAGAIN;
    }
    if (b->fill >= 0 && b->host_len[b->fill]) close_fill(b);
    if (!b->pin_len) b->pin_ptr = data;
    b->pin_len += len; *taken = len;                     // (all of it is the pass's from here on: it leaves for the device as buffers fall free -- now, and inside bgp_batch_next_window)
    BGP_TRY(ship_ready(c, b));
    return BGP_OK;
    BGP_API_CATCH(c)
}

/* Zero-copy staging: the caller (an inflater, a reader) writes its bytes straight into the pinned buffer the next copy to HBM
 * starts from -- the shape of decompressBgzfBlock writing into the block's own buffer (block.d:127-216) -- instead of handing them
 * to bgp_batch_push to be copied there. */
int bgp_batch_acquire(bgp_ctx* c, uint8_t** ptr, size_t* cap) {
    BGP_API_TRY
    if (!c || !ptr || !cap) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b || b->ended || b->feed) return BGP_ESTATE;      // (a pass bgp_stream_bam_begin feeds from a file takes no bytes from the caller)
    if (b->acq >= 0) return BGP_ESTATE;                  // one acquisition at a time: commit (0 bytes if need be) before the next
    BGP_HIP(hipSetDevice(c->device));
    *ptr = nullptr; *cap = 0;
    if (b->pin_len) { BGP_TRY(ship_ready(c, b)); if (b->pin_len) return BGP_EAGAIN; }      // (page-locked bytes pushed earlier go first)
    // a buffer with less room than one BGZF block inflates to (64 KiB, constants.d:26-61) is shipped as it is: a caller that writes
    // whole blocks could otherwise be handed a tail it can never use, with no way to move on
    { const int rc = fill_ready(c, b, true); if (rc != BGP_OK) return rc; }      // BGP_EAGAIN: hand out windows first
    const int h = b->fill;
    *ptr = b->host[h].as<uint8_t>() + b->host_len[h]; *cap = (size_t)(b->batch_bytes - b->host_len[h]);
    b->acq = h; b->acq_off = b->host_len[h];
    return BGP_OK;
    BGP_API_CATCH(c)
}
int bgp_batch_commit(bgp_ctx* c, size_t n) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b || b->ended || b->feed) return BGP_ESTATE;      // (a pass bgp_stream_bam_begin feeds from a file takes no bytes from the caller)
    if (b->acq < 0) return BGP_ESTATE;                   // nothing was acquired
    const int h = b->acq;                                // the buffer that was handed out, whatever `fill` says by now
    if (b->host_len[h] != b->acq_off || n > b->batch_bytes - b->host_len[h]) return BGP_EINVAL;
    b->host_len[h] += n;
    b->acq = -1;
    BGP_HIP(hipSetDevice(c->device));
    BGP_TRY(ship_if_full(c, b));
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_batch_set_md(bgp_ctx* c, int use_md_tag, int skip_zero_coverage) {
    BGP_API_TRY
    if (!c) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b || b->any_batch) return BGP_ESTATE;               // right after bgp_batch_begin
    c->batch_md = use_md_tag != 0; c->batch_md_skip_zero = skip_zero_coverage != 0;
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_batch_end(bgp_ctx* c) {
    BGP_API_TRY
    if (!c || !c->batch) return BGP_ESTATE;
    if (c->batch->acq >= 0 || c->batch->feed) return BGP_ESTATE;           // an acquired buffer was never committed | the file says when the stream is over
    c->batch->ended = true;
    return BGP_OK;
    BGP_API_CATCH(c)
}

int bgp_batch_next_window(bgp_ctx* c, bgp_columns* o) {
    BGP_API_TRY
    if (!c || !o) return BGP_EINVAL;
    Batch* b = c->batch;
    if (!b) return BGP_ESTATE;
    BGP_HIP(hipSetDevice(c->device));
    if (b->feed && b->cur < 0) BGP_TRY(feed_pump(c, b, false));          // (a file feeds the pass: its first batches leave as soon as the file has been read that far; later ones when a batch's kernels are through, process_next)
    for (;;) {
        if (b->cur >= 0) {                                               // hand out the current batch's final columns, run after run
            while (b->run < c->runs.size()) {
                const bgp_run& r = c->runs[b->run];
                uint64_t lo = r.start_pos, hi = r.end_pos;
                if (b->resume && b->run == 0 && r.ref_id == b->resume_ref) lo = std::max(lo, b->resume_pos);
                if (b->cut && b->run + 1 == c->runs.size() && r.ref_id == b->cut_ref) hi = std::min(hi, b->cut_pos);
                if (!b->run_open) { b->next_pos = lo; b->run_open = true; }
                if (b->next_pos >= hi) { ++b->run; b->run_open = false; continue; }
                const uint64_t a = b->next_pos, e = std::min(hi, a + b->window);
                if (c->batch_maq) { c->it.maq = 1; c->it.maq_depcorr = c->batch_maq_depcorr; c->it.maq_eta = c->batch_maq_eta; c->it.maq_min_bq = c->batch_maq_min_bq; }
                BGP_TRY(emit_window(c, b->run, a, e, b->want_entries, c->batch_md, 0, o));
                if (b->want_entries && b->ord_base + c->n_records > (1ull << 32)) {     // bgp_entry.read_ordinal is 32 bits wide
                    c->last_error = "batched staging with per-read entries: the stream has more than 2^32 records (bgp_entry.read_ordinal would wrap)";
                    return BGP_EINVAL;
                }
                if (b->want_entries && b->ord_base) {                    // stream-wide read ordinals
                    bgp_entry* en = c->h_entries.as<bgp_entry>();
                    const uint64_t ne = o->entry_off[o->n_positions];
                    for (uint64_t k = 0; k < ne; ++k) en[k].read_ordinal += (uint32_t)b->ord_base;
                }
                b->next_pos = e;
                return BGP_OK;
            }
            if (b->cut) { b->resume = true; b->resume_ref = b->cut_ref; b->resume_pos = b->cut_pos; }      // (a batch without covering reads leaves the mark where it was)
            if (b->cur_final) return BGP_EOF;
        }
        // the next batch
        if (b->n_queue == 0 && b->feed) BGP_TRY(feed_pump(c, b, true));       // (waits for the file: a batch is queued, or the stream is over)
        if (b->n_queue == 0 && b->pin_len) BGP_TRY(ship_ready(c, b));           // (the caller's page-locked bytes still waiting: the buffer the last batch gave up is theirs)
        if (b->n_queue == 0) {
            if (!b->ended) return b->feed ? BGP_ESTATE : BGP_EAGAIN;        // push more, or say the stream is over
            if (!b->n_full) {
                if (b->fill < 0) next_fill(b);
                if ((b->fill >= 0 && b->host_len[b->fill]) || !b->any_batch) close_fill(b);      // what is left in the host buffer is the last batch (an empty stream is one, too)
                else if (b->cur >= 0 && !b->cur_final) close_fill(b);                            // the stream ended exactly on a batch: what was carried is the last batch
                else return BGP_EOF;
            }
            const int rc = ship(c, b);
            if (rc != BGP_OK) return rc == BGP_EAGAIN ? BGP_ESTATE : rc;
            if (!b->n_queue) return BGP_ESTATE;
        }
        BGP_TRY(process_next(c, b));
    }
    BGP_API_CATCH(c)
}

/* record ordinal of the current batch's first record in the whole stream (bgp_decode_fetch's tables are per batch) */
int bgp_batch_ordinal_base(bgp_ctx* c, uint64_t* base) {
    BGP_API_TRY
    if (!c || !c->batch || !base) return BGP_ESTATE;
    *base = c->batch->ord_base;
    return BGP_OK;
    BGP_API_CATCH(c)
}

}  // extern "C"
