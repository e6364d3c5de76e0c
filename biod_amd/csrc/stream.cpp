// stream.cpp -- a BAM FILE through the device in bounded memory: bgp_stream_bam_begin + bgp_batch_next_window.
//
// Reference shape: BioD never holds the file.  BgzfInputStream keeps a ring of 2 x pool.size blocks in flight -- read, handed to a
// TaskPool task that inflates them (decompressBgzfBlock), consumed in file order (bio/core/bgzf/inputstream.d:386-447,455-483,
// block.d:127-216) -- under a pull-driven consumer (BamReadRange -> PileupRange).  Here the ring is three rings, and the consumer's pull
// is bgp_batch_next_window:
//   * reader threads pread the file in 8 MiB chunks (+ 192 KiB of overlap, so a block that starts in a chunk lies whole in its buffer)
//     into a pinned buffer each, walk the block headers in file order (every header names the next), and copy the chunk to its slot of
//     a ring of compressed chunks in HBM; a slot is taken again only when every block that starts in its old chunk has been inflated;
//   * the consumer's thread cuts the list of blocks into batches of at most batch_bytes inflated bytes, and for each -- as soon as one of
//     the batched pass's two device buffers is free -- enqueues k_inflate (inflate.hip: a wave a block, CRC checked) from the ring into
//     that buffer, behind the uploads' events, on one of the ctx's two inflate streams: the inflate of batch k + 1 runs beside the
//     kernels of batch k, and no inflated byte ever exists on the host;
//   * from there on a batch is batch.cpp's: carry in front, framing, record table, pileup, the cut, windows out.
// Device memory: two batch buffers, the compressed ring (batch_bytes and a few chunks), the tables and columns of ONE batch -- whatever
// the size of the file.  The columns (and entries, and reference bases with bgp_batch_set_md) are those of the one-shot load: tested.
#include "batch.h"

#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

struct FileFeed {
    static constexpr uint64_t OV = 192ull << 10;
    uint64_t CH = 8ull << 20, STRIDE = CH + OV;             // a chunk of the file, and a slot of the rings: the chunk and the overlap (BGP_STREAM_CHUNK: small chunks, so that test files of a few MB go round the ring many times)
    int fd = -1; uint64_t fsz = 0, n_chunks = 0;
    uint64_t q = 0;                                        // the first record's offset in the inflated stream (behind the BAM header)
    int device = 0; hipStream_t io_up = nullptr;
    int T = 1; uint64_t ND = 0;                            // reader threads; slots of the device ring
    HostBuf h_ring; DevBuf d_ring;
    std::vector<hipEvent_t> ev_up, ev_free;                // per slot: its chunk is in HBM | every block of its chunk has been inflated
    std::vector<std::thread> th;
    std::mutex mu, up_mu; std::condition_variable cv;
    // under mu
    uint64_t turn = 0, scan_off = 0, scan_uoff = 0; int scan_rc = BGP_OK; std::string scan_err;
    std::vector<BgzfBlk> fresh;                            // blocks scanned and not yet taken by the consumer
    std::vector<uint8_t> uploaded, released;               // per chunk
    bool stop = false;
    std::atomic<uint64_t> next{0}; std::atomic<int> hip_rc{BGP_OK};
    // the consumer's side
    std::deque<BgzfBlk> todo;                              // blocks not yet in a batch, file order
    uint64_t seen_turn = 0, rel_next = 0;                  // chunks scanned as of the last look; chunks [0, rel_next) are released
    uint64_t u_next = 0;                                   // inflated offset the next batch begins at (q for the first)
    bool all_dispatched = false; int n_batch = 0, prev_q = -1;
    uint64_t slots = 0;                                    // BGZF blocks the device inflates at a time: a wave each, 8 waves a SIMD (inflate.hip)
    bool verbose = getenv("BGP_VERBOSE") != nullptr; std::chrono::steady_clock::time_point t_begin = std::chrono::steady_clock::now();
    double ms_now() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); }
    DevBuf d_desc[Batch::NDV]; HostBuf h_desc[Batch::NDV], h_status[Batch::NDV]; uint32_t n_desc[Batch::NDV] = {0, 0, 0, 0};
    uint64_t first_coff[Batch::NDV] = {0, 0, 0, 0};

    ~FileFeed() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        next.store(~0ull >> 1);
        cv.notify_all();
        for (auto& x : th) if (x.joinable()) x.join();
        if (io_up) (void)hipStreamSynchronize(io_up);
        for (hipEvent_t e : ev_up) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_free) if (e) (void)hipEventDestroy(e);
        h_ring.release(); d_ring.release();
        for (int k = 0; k < Batch::NDV; ++k) { d_desc[k].release(); h_desc[k].release(); h_status[k].release(); }
        if (fd >= 0) close(fd);
    }
};

namespace {

void reader(FileFeed* f, int t) {
    if (hipSetDevice(f->device) != hipSuccess) f->hip_rc.store(BGP_EDEVICE);
    uint8_t* buf = f->h_ring.as<uint8_t>() + (size_t)t * f->STRIDE;
    int64_t last = -1;                                     // the chunk whose upload reads this buffer
    for (;;) {
        const uint64_t k = f->next.fetch_add(1);
        if (k >= f->n_chunks) break;
        if (k >= f->ND) {                                  // the slot's old chunk: inflated?
            std::unique_lock<std::mutex> lk(f->mu);
            f->cv.wait(lk, [&] { return f->stop || f->released[k - f->ND]; });
            if (f->stop) break;
            lk.unlock();
            if (hipEventSynchronize(f->ev_free[k % f->ND]) != hipSuccess) f->hip_rc.store(BGP_EDEVICE);
        }
        if (last >= 0 && hipEventSynchronize(f->ev_up[(uint64_t)last % f->ND]) != hipSuccess) f->hip_rc.store(BGP_EDEVICE);      // the previous piece has left this buffer
        const uint64_t lo = k * f->CH, hi = std::min(f->fsz, lo + f->STRIDE);
        uint64_t got = 0;
        while (got < hi - lo) { const ssize_t r = pread(f->fd, buf + got, hi - lo - got, (off_t)(lo + got)); if (r <= 0) break; got += (uint64_t)r; }
        // (the chunk's blocks are made known to the consumer below, together with the turn -- behind the upload's event: a batch cut between
        // the scan and the upload would have its k_inflate wait for an event not yet recorded, i.e. not at all, and read the slot's old bytes)
        std::vector<BgzfBlk> mine;
        {
            std::unique_lock<std::mutex> lk(f->mu);
            f->cv.wait(lk, [&] { return f->stop || f->turn == k; });
            if (f->stop) break;
            if (f->scan_rc == BGP_OK) {
                if (got < hi - lo) { f->scan_rc = BGP_ETRUNC; f->scan_err = "short read"; }
                else if (f->scan_off < lo) { f->scan_rc = BGP_EINVAL; f->scan_err = "BGZF block larger than 64 KB"; }
                else {
                    try { f->scan_rc = bgzf_scan_blocks(buf - lo, hi, std::min(f->fsz, lo + f->CH), f->scan_off, f->scan_uoff, mine, f->scan_err); }
                    catch (...) { f->scan_rc = BGP_ENOMEM; f->scan_err = "out of host memory while reading the file"; }
                }
            }
        }
        if (f->hip_rc.load() == BGP_OK) {
            std::lock_guard<std::mutex> ul(f->up_mu);      // (a copy and its event are enqueued together)
            if (hipMemcpyAsync(f->d_ring.as<uint8_t>() + (k % f->ND) * f->STRIDE, buf, hi - lo, hipMemcpyHostToDevice, f->io_up) != hipSuccess
                || hipEventRecord(f->ev_up[k % f->ND], f->io_up) != hipSuccess) f->hip_rc.store(BGP_EDEVICE);
            else last = (int64_t)k;
        }
        {   // (the turn moves on only now: a chunk the consumer sees as scanned has its upload's event recorded)
            std::lock_guard<std::mutex> lk(f->mu);
            try { f->fresh.insert(f->fresh.end(), mine.begin(), mine.end()); }
            catch (...) { if (f->scan_rc == BGP_OK) { f->scan_rc = BGP_ENOMEM; f->scan_err = "out of host memory while reading the file"; } }
            f->uploaded[k] = 1; f->turn = k + 1;
        }
        f->cv.notify_all();
    }
    if (last >= 0) (void)hipEventSynchronize(f->ev_up[(uint64_t)last % f->ND]);
}

}  // namespace

void feed_release(bgp_ctx*, Batch* b) {
    delete b->feed;
    b->feed = nullptr;
}

// the inflate's verdict on the batch in dev[q] (its event has passed: the status words are in pinned memory)
int feed_check(bgp_ctx* c, Batch* b, int q) {
    FileFeed* f = b->feed;
    const double w0 = f->verbose ? f->ms_now() : 0;
    BGP_HIP(hipEventSynchronize(b->loaded_ev[q]));
    if (f->verbose) fprintf(stderr, "[biodgpu]   %8.2f ms: buffer %d inflated (waited %.2f ms for it)\n", f->ms_now(), q, f->ms_now() - w0);
    const int32_t* st = f->h_status[q].as<int32_t>();
    const bgp_inflate_block* d = f->h_desc[q].as<bgp_inflate_block>();
    for (uint32_t i = 0; i < f->n_desc[q]; ++i)
        if (st[i]) { c->last_error = "zlib: corrupted BGZF block (device inflate, block " + std::to_string(i) + " of the batch that begins at file offset " + std::to_string(f->first_coff[q]) + ", " + std::to_string(d[i].isize) + " bytes, reason " + std::to_string(st[i]) + ")"; return BGP_EINVAL; }
    return BGP_OK;
}

int feed_pump(bgp_ctx* c, Batch* b, bool need_one) {
    FileFeed* f = b->feed;
    for (;;) {
        if (f->all_dispatched) { b->ended = true; return BGP_OK; }
        int q = -1;
        for (int k = 0; k < b->n_dev; ++k) if (b->state[k] == Batch::FREE) { q = k; break; }
        if (q < 0) return BGP_OK;                          // (both device buffers taken: the consumer hands out windows first)
        const bool must = need_one && b->n_queue == 0;
        // (the first batch is a small one: its kernels and windows start while the second, full-size batch inflates -- 40 ms to the first
        // window instead of 90 with batches of 1 GiB)
        const uint64_t batch_cap = f->n_batch == 0 ? std::min<uint64_t>(b->batch_bytes, 256ull << 20) : b->batch_bytes;
        // what has been scanned since the last look; enough for a batch?
        bool scanned_all = false, ring_full = false; int src = BGP_OK;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(f->mu);
                if (!f->fresh.empty()) { f->todo.insert(f->todo.end(), f->fresh.begin(), f->fresh.end()); f->fresh.clear(); }
                f->seen_turn = f->turn; src = f->scan_rc;
                scanned_all = f->seen_turn >= f->n_chunks || src != BGP_OK;
                // (the ring is sized for blocks that are no larger than what they inflate to.  A file of tiny or empty blocks -- 28 bytes of
                // container for nothing -- fills it before a batch's worth of bytes has been seen: the readers then wait for a slot, and a
                // slot is only given back behind a batch.  What has been scanned is the batch.)
                uint64_t bytes = 0;
                for (const BgzfBlk& x : f->todo) { if (x.uoff + x.isize > f->q) bytes += x.isize; if (bytes + 65536 > batch_cap) break; }
                ring_full = !scanned_all && bytes + 65536 <= batch_cap && f->seen_turn >= f->rel_next + f->ND;
                if (scanned_all || ring_full || bytes + 65536 > batch_cap || f->hip_rc.load() != BGP_OK) break;
                if (!must) return BGP_OK;                  // (not yet: looked at again at the next call)
                f->cv.wait_for(lk, std::chrono::milliseconds(2));
            }
        }
        if (f->hip_rc.load() != BGP_OK) { c->last_error = "HIP error while uploading the compressed file"; return f->hip_rc.load(); }
        // the batch: blocks of todo while their inflated bytes fit
        std::vector<bgp_inflate_block> desc;
        uint64_t bytes = 0, c_first = ~0ull, c_last = 0; size_t n_take = 0;
        const uint64_t u0 = f->n_batch == 0 ? f->q : f->u_next;
        constexpr uint64_t SLACK = 65536;                  // (the block that holds the first record starts up to 64 KB before it: b->reserve >= SLACK)
        // (a wave inflates a block and the device holds f->slots of them at a time: a batch of a few hundred blocks more than a multiple of
        // that runs a whole extra round for them -- 16 777 blocks of 64 000 bytes, 1 GiB, took three block-times where 16 384 take two: 29
        // against ~21 ms.  A batch of more than one round's worth is cut at a multiple of the round.)
        size_t n_inf = 0;
        for (const BgzfBlk& x : f->todo) { if (x.isize && x.uoff + x.isize > f->q) ++n_inf; }
        uint64_t fit = 0, fit_bytes = 0;
        for (const BgzfBlk& x : f->todo) { if (x.isize && x.uoff + x.isize > f->q) { if (fit_bytes + x.isize > batch_cap && fit) break; fit_bytes += x.isize; ++fit; } }
        const uint64_t blk_cap = (f->slots && fit > f->slots && !(scanned_all && fit == n_inf) && !ring_full) ? fit / f->slots * f->slots : ~0ull;
        for (const BgzfBlk& x : f->todo) {
            if (x.isize && x.uoff + x.isize > f->q) {
                if ((bytes + x.isize > batch_cap && bytes) || desc.size() >= blk_cap) break;
                const uint64_t k = x.coff / f->CH;
                desc.push_back(bgp_inflate_block{(k % f->ND) * f->STRIDE + (x.coff - k * f->CH) + x.cdata_off, SLACK + x.uoff - u0, x.cdata_len, x.isize, x.crc, 0});
                bytes += x.uoff >= u0 ? x.isize : x.uoff + x.isize - u0;
                c_first = std::min(c_first, k); c_last = std::max(c_last, k);
            }
            ++n_take;
        }
        const bool last_batch = scanned_all && n_take == f->todo.size();
        // (with the default two device buffers one inflate runs at a time, on the first inflate stream: the second is the window prefetch's
        // -- api.cpp: emit_window -- and a window's copy queued behind k_inflate there waited for the whole inflate, 9 GB/s of columns)
        hipStream_t is = c->io_inf[b->n_dev > 2 ? (f->n_batch & 1) : 0];
        // slots whose chunks have no block left to inflate go back to the readers: an event on the inflate stream, behind every inflate that
        // reads them
        auto release_slots = [&]() -> int {
            const uint64_t upto = f->todo.empty() ? f->seen_turn : std::min<uint64_t>(f->seen_turn, f->todo.front().coff / f->CH);
            if (upto > f->rel_next) {
                for (uint64_t k = f->rel_next; k < upto; ++k) BGP_HIP(hipEventRecord(f->ev_free[k % f->ND], is));
                { std::lock_guard<std::mutex> lk(f->mu); for (uint64_t k = f->rel_next; k < upto; ++k) f->released[k] = 1; }
                f->rel_next = upto;
                f->cv.notify_all();
            }
            return BGP_OK;
        };
        if (desc.empty() && !last_batch) {      // (header blocks, empty blocks: nothing to inflate -- but their chunks' slots are wanted back)
            for (size_t i = 0; i < n_take; ++i) f->todo.pop_front();
            if (f->prev_q >= 0) BGP_HIP(hipStreamWaitEvent(is, b->loaded_ev[f->prev_q], 0));
            BGP_TRY(release_slots());
            continue;
        }
        if (desc.empty() && last_batch) {
            for (size_t i = 0; i < n_take; ++i) f->todo.pop_front();
            if (src != BGP_OK) { c->last_error = f->scan_err; return src; }      // (everything in front of the bad block has been handed on)
            f->all_dispatched = true; b->ended = true;
            return BGP_OK;
        }
        // enqueue: descriptors, k_inflate behind the uploads of the chunks it reads, the status back, the batch's event
        const uint32_t nd = (uint32_t)desc.size();
        BGP_TRY(b->dev[q].reserve(b->reserve + b->batch_bytes + 256));
        BGP_TRY(f->d_desc[q].reserve((size_t)nd * (sizeof(bgp_inflate_block) + 4) + 64));
        BGP_TRY(f->h_desc[q].reserve((size_t)nd * sizeof(bgp_inflate_block) + 64)); BGP_TRY(f->h_status[q].reserve((size_t)nd * 4 + 64));
        memcpy(f->h_desc[q].p, desc.data(), (size_t)nd * sizeof(bgp_inflate_block));
        f->n_desc[q] = nd; f->first_coff[q] = f->todo.front().coff;
        for (uint64_t k = c_first; k <= c_last; ++k) BGP_HIP(hipStreamWaitEvent(is, f->ev_up[k % f->ND], 0));
        int32_t* d_status = (int32_t*)(f->d_desc[q].as<uint8_t>() + (size_t)nd * sizeof(bgp_inflate_block));
        BGP_HIP(hipMemcpyAsync(f->d_desc[q].p, f->h_desc[q].p, (size_t)nd * sizeof(bgp_inflate_block), hipMemcpyHostToDevice, is));
        BGP_TRY(inflate_blocks_device(c, f->d_ring.as<uint8_t>(), b->dev[q].as<uint8_t>() + b->reserve - SLACK, f->d_desc[q].as<bgp_inflate_block>(), nd, d_status, bytes, is));
        BGP_HIP(hipMemcpyAsync(f->h_status[q].p, d_status, (size_t)nd * 4, hipMemcpyDeviceToHost, is));
        BGP_HIP(hipEventRecord(b->loaded_ev[q], is));
        // (a chunk's blocks can lie in several batches, the batches on the two inflate streams in turn: what follows on this stream -- the
        // events that hand slots back to the readers -- also waits for the batch before this one, on the other stream)
        if (f->prev_q >= 0) BGP_HIP(hipStreamWaitEvent(is, b->loaded_ev[f->prev_q], 0));
        f->prev_q = q;
        b->new_off[q] = b->reserve; b->new_len[q] = bytes;
        b->state[q] = Batch::LOADED; b->queue[b->n_queue++] = q;
        if (f->verbose) fprintf(stderr, "[biodgpu]   %8.2f ms: batch %d -> buffer %d on inflate stream %d: %u blocks, %.1f MB, chunks %llu..%llu (scanned %llu of %llu), %d queued%s\n", f->ms_now(), f->n_batch, q, is == c->io_inf[0] ? 0 : 1, nd, bytes / 1e6,
                                (unsigned long long)c_first, (unsigned long long)c_last, (unsigned long long)f->seen_turn, (unsigned long long)f->n_chunks, b->n_queue, ring_full ? " -- cut because the ring was full" : "");
        f->u_next = u0 + bytes; ++f->n_batch;
        for (size_t i = 0; i < n_take; ++i) f->todo.pop_front();
        BGP_TRY(release_slots());
        if (last_batch) {
            if (src != BGP_OK) return BGP_OK;              // (the scan's error is reported when the consumer asks for what lies behind this batch: the next call finds nothing left to inflate)
            f->all_dispatched = true; b->ended = true;
        }
    }
}

extern "C" {

int bgp_stream_bam_begin(bgp_ctx* c, const char* path, uint64_t batch_bytes, int want_entries, uint32_t window_positions) {
    BGP_API_TRY
    if (!c || !path) return BGP_EINVAL;
    BGP_HIP(hipSetDevice(c->device));
    std::unique_ptr<FileFeed> f(new FileFeed());
    f->fd = open(path, O_RDONLY);
    if (f->fd < 0) { c->last_error = std::string("cannot open ") + path; return BGP_EINVAL; }
    const int fd = f->fd;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { c->last_error = "fstat failed"; return BGP_EINVAL; }
    f->fsz = (uint64_t)sb.st_size;
    std::vector<RefInfo> refs; uint64_t q = 0; std::string err;
    { const int rc = bgzf_read_bam_header(fd, f->fsz, refs, &q, err); if (rc < 0) { c->last_error = err; return rc; } }
    // (a batch holds whole BGZF blocks: at least two of them.  bgp_reset inside: the references are set behind it.  No pinned buffers
    // for pushed bytes: nobody pushes)
    BGP_TRY(batch_begin_core(c, std::max<uint64_t>(batch_bytes ? batch_bytes : (1ull << 30), 128ull << 10), want_entries, window_positions, false));
    Batch* b = c->batch;
    c->refs = refs; c->n_ref_hint = (int32_t)refs.size();
    f->q = q; f->device = c->device; f->io_up = c->io_up;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) f->slots = (uint64_t)prop.multiProcessorCount * 32; else (void)hipGetLastError(); }
    if (const char* e = getenv("BGP_STREAM_ROUND")) f->slots = (uint64_t)atoll(e);      // (0: batches are cut by bytes alone)
    if (const char* e = getenv("BGP_STREAM_CHUNK")) { f->CH = std::max<uint64_t>(65536, (uint64_t)atoll(e)); f->STRIDE = f->CH + FileFeed::OV; }
    f->n_chunks = (f->fsz + f->CH - 1) / f->CH;
    const uint64_t t_max = getenv("BGP_READ_THREADS") ? (uint64_t)atoi(getenv("BGP_READ_THREADS")) : 8ull;
    f->T = (int)std::max<uint64_t>(1, std::min<uint64_t>({t_max, (uint64_t)bgp_default_threads(), f->n_chunks}));
    // the ring holds the blocks of the batches that can be inflating at a time (every device buffer but the current batch's) whatever they compress to (a BGZF block is never much larger than what it inflates to: 26 bytes of
    // header and footer, 5 of a stored deflate block), a chunk cut at either end of it, and one chunk a reader ahead
    f->ND = std::min<uint64_t>(f->n_chunks, ((uint64_t)(b->n_dev - 1) * (b->batch_bytes + b->batch_bytes / 512) + f->CH - 1) / f->CH + 3 + (uint64_t)f->T);
    f->ND = std::max<uint64_t>(f->ND, 1);
    int rc = f->d_ring.reserve((size_t)(f->ND * f->STRIDE + 256));
    if (rc == BGP_OK) rc = f->h_ring.reserve((size_t)f->T * f->STRIDE);
    if (rc != BGP_OK) { batch_release(c); return rc; }
    f->ev_up.assign(f->ND, nullptr); f->ev_free.assign(f->ND, nullptr);
    for (uint64_t k = 0; k < f->ND; ++k)
        if (hipEventCreateWithFlags(&f->ev_up[k], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&f->ev_free[k], hipEventDisableTiming) != hipSuccess) { c->last_error = "hipEventCreate failed"; batch_release(c); return BGP_EDEVICE; }
    f->uploaded.assign(f->n_chunks, 0); f->released.assign(f->n_chunks, 0);
    FileFeed* fp = f.release();
    b->feed = fp;
    try { for (int t = 0; t < fp->T; ++t) fp->th.emplace_back(reader, fp, t); }
    catch (...) { c->last_error = "cannot start the reader threads"; batch_release(c); return BGP_ENOMEM; }
    return BGP_OK;
    BGP_API_CATCH(c)
}

}  // extern "C"
