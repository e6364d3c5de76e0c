// bgzf_host.cpp -- host feeder H1: BGZF container -> inflated record bytes staged for the device.
//
// Mirrors the container layer BioD runs below the hot path: block header / BC subfield parse
// (bio/core/bgzf/inputstream.d:54-199), raw-deflate inflate with size + crc32 checks
// (bio/core/bgzf/block.d:127-216), fanned out over worker threads in file order like
// BgzfInputStream's TaskPool ring (inputstream.d:386-483), then the BAM magic, header text and
// reference table (bio/std/hts/bam/reader.d:111-116,579-597; referenceinfo.d).
// The same third-party arithmetic as the reference: system zlib (bio/core/utils/zlib.d:6-245).
#include "common.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

namespace {

typedef BgzfBlk Blk;      // (common.h: shared with the file feeder, stream.cpp)

// Block headers from file offset `off` on, while off < stop; f[x] is readable for x < n (n = file size, or the end
// of the bytes read so far -- a block is at most 64 KB, so a reader that holds 128 KB past `stop` sees it whole).
int scan_blocks(const uint8_t* f, uint64_t n, uint64_t stop, uint64_t& off, uint64_t& uoff, std::vector<Blk>& out, std::string& err) {
    while (off < stop) {
        if (n - off < 4) break;                                          // inputstream.d:76-84: short magic read -> end
        if (!(f[off] == 0x1F && f[off + 1] == 0x8B && f[off + 2] == 8 && f[off + 3] == 4)) { err = "wrong BGZF magic"; return BGP_EINVAL; }
        if (n - off < 12) { err = "truncated BGZF header"; return BGP_ETRUNC; }
        const uint32_t xlen = f[off + 10] | (f[off + 11] << 8);
        uint64_t p = off + 12; const uint64_t xend = p + xlen;
        if (xend > n) { err = "truncated BGZF extra field"; return BGP_ETRUNC; }
        int bsize = -1; uint32_t len = 0;
        while (len < xlen) {                                             // inputstream.d:107-152
            if (p + 4 > n) { err = "truncated BGZF subfield"; return BGP_ETRUNC; }
            const uint8_t si1 = f[p], si2 = f[p + 1]; const uint32_t slen = f[p + 2] | (f[p + 3] << 8);
            if (si1 == 66 && si2 == 67) {
                if (slen != 2) { err = "wrong BC subfield length"; return BGP_EINVAL; }
                if (bsize >= 0) { err = "duplicate field with block size"; return BGP_EINVAL; }
                if (p + 6 > n) { err = "truncated BC subfield"; return BGP_ETRUNC; }
                bsize = f[p + 4] | (f[p + 5] << 8);
            }
            p += 4 + slen; len += 4 + slen;
        }
        if (len != xlen) { err = "subfield lengths do not add up to gzip_extra_length"; return BGP_EINVAL; }
        if (bsize < 0) { err = "block size was not found in any subfield"; return BGP_EINVAL; }
        const int64_t cdata = (int64_t)bsize - xlen - 19;
        if (cdata < 0 || cdata > 65536) { err = "bad compressed data size"; return BGP_EINVAL; }
        if (xend + (uint64_t)cdata + 8 > n) { err = "truncated BGZF block"; return BGP_ETRUNC; }
        Blk b; b.coff = off; b.cdata_off = (uint32_t)(xend - off); b.cdata_len = (uint32_t)cdata;
        memcpy(&b.crc, f + xend + cdata, 4); memcpy(&b.isize, f + xend + cdata + 4, 4);
        if (b.isize > 65536) { err = "uncompressed block size must be within 65536 bytes"; return BGP_EINVAL; }
        b.uoff = uoff; uoff += b.isize;
        out.push_back(b);
        off = xend + (uint64_t)cdata + 8;
    }
    return BGP_OK;
}

int parse_blocks(const uint8_t* f, uint64_t n, std::vector<Blk>& out, std::string& err) {
    uint64_t off = 0, uoff = 0;
    return scan_blocks(f, n, n, off, uoff, out, err);
}

int inflate_cdata(const uint8_t* cdata, const Blk& b, uint8_t* dst) {     // block.d:127-216; cdata = the block's compressed payload
    if (b.isize == 0) return BGP_OK;
    z_stream zs; memset(&zs, 0, sizeof zs);
    zs.next_in = const_cast<Bytef*>(cdata); zs.avail_in = b.cdata_len;
    if (inflateInit2(&zs, -15) != Z_OK) return BGP_ENOMEM;
    zs.next_out = dst; zs.avail_out = b.isize;
    const int e = inflate(&zs, Z_FINISH);
    const uLong produced = zs.total_out;
    inflateEnd(&zs);
    if (e != Z_STREAM_END || produced != b.isize) return BGP_EINVAL;
    if ((uint32_t)crc32(0, dst, b.isize) != b.crc) return BGP_EINVAL;
    return BGP_OK;
}
int inflate_block(const uint8_t* f, const Blk& b, uint8_t* dst) { return inflate_cdata(f + b.coff + b.cdata_off, b, dst); }   // f = the whole file

}  // namespace

// the same scan and the same zlib call for the feeder of bgp_stream_bam_begin (stream.cpp)
int bgzf_scan_blocks(const uint8_t* f, uint64_t n, uint64_t stop, uint64_t& off, uint64_t& uoff, std::vector<BgzfBlk>& out, std::string& err) { return scan_blocks(f, n, stop, off, uoff, out, err); }
int bgzf_inflate_cdata(const uint8_t* cdata, const BgzfBlk& b, uint8_t* dst) { return inflate_cdata(cdata, b, dst); }

// Worker threads when the caller passes 0: the CPUs this process may run on (affinity mask), cut by the cgroup
// CPU quota -- a container that shows 256 logical CPUs under a 16-CPU cpu.max inflates slower on 256 threads.
int bgp_default_threads() {
    int n = (int)std::thread::hardware_concurrency();
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n > 0 ? n : CPU_COUNT(&set), CPU_COUNT(&set));
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32]; long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, 2 * atoll(q) / period));
        fclose(f);
    }
    return n < 1 ? 1 : n;
}

// BAM magic + header text + reference table (reader.d:111-116,579-597) from the first `avail` inflated bytes.
// Returns 1 and the offset of the first record when complete, 0 when more bytes are needed, <0 on error.
int bgp_parse_bam_header(const uint8_t* p, uint64_t avail, std::vector<RefInfo>& refs, uint64_t* records_off, std::string& err) {
    if (avail < 12) return 0;
    if (memcmp(p, "BAM\1", 4) != 0) { err = "wrong BAM magic"; return BGP_EINVAL; }
    int32_t l_text; memcpy(&l_text, p + 4, 4);
    if (l_text < 0) { err = "negative header length"; return BGP_EINVAL; }
    uint64_t q = 8 + (uint64_t)(uint32_t)l_text;
    if (q + 4 > avail) return 0;
    int32_t n_ref; memcpy(&n_ref, p + q, 4); q += 4;
    if (n_ref < 0) { err = "negative reference count"; return BGP_EINVAL; }
    refs.clear();
    for (int32_t i = 0; i < n_ref; ++i) {
        if (q + 4 > avail) return 0;
        int32_t l_name; memcpy(&l_name, p + q, 4); q += 4;
        if (l_name < 1) { err = "bad reference name length"; return BGP_EINVAL; }
        if (q + (uint64_t)l_name + 4 > avail) return 0;
        RefInfo ri; ri.name.assign((const char*)p + q, (size_t)l_name - 1); q += (uint64_t)l_name;
        memcpy(&ri.length, p + q, 4); q += 4;
        refs.push_back(ri);
    }
    *records_off = q;
    return 1;
}

int bgzf_load_bam(bgp_ctx* c, const char* path, int n_threads) {
    const bool verbose = getenv("BGP_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { c->last_error = std::string("cannot open ") + path; return BGP_EINVAL; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); c->last_error = "fstat failed"; return BGP_EINVAL; }
    const size_t fsz = (size_t)sb.st_size;
    const uint8_t* file = fsz ? (const uint8_t*)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    struct CloseFd { int fd; ~CloseFd() { close(fd); } } close_fd{fd};
    if (fsz && file == (const uint8_t*)MAP_FAILED) { c->last_error = "mmap failed"; return BGP_EINVAL; }
    struct Unmap { const uint8_t* p; size_t n; ~Unmap() { if (p && n) munmap((void*)p, n); } } unmap{file, fsz};

    std::vector<Blk> blocks;
    int rc = parse_blocks(file, fsz, blocks, c->last_error);
    if (rc < 0) return rc;
    const uint64_t total = blocks.empty() ? 0 : blocks.back().uoff + blocks.back().isize;
    const auto t1 = now();

    // the header lives in the first block(s): inflate serially until it parses
    std::vector<uint8_t> head;
    uint64_t q = 0; size_t jh = 0; int got = 0;
    while (jh < blocks.size() && got == 0) {
        head.resize((size_t)(blocks[jh].uoff + blocks[jh].isize));
        rc = inflate_block(file, blocks[jh], head.data() + blocks[jh].uoff);
        if (rc < 0) { c->last_error = "zlib: corrupted BGZF block"; return rc; }
        ++jh;
        got = bgp_parse_bam_header(head.data(), head.size(), c->refs, &q, c->last_error);
        if (got < 0) return got;
    }
    if (got == 0) { c->last_error = total < 4 ? "wrong BAM magic" : "truncated BAM header"; return total < 4 ? BGP_EINVAL : BGP_ETRUNC; }
    c->n_ref_hint = (int32_t)c->refs.size();

    // every worker inflates straight into the pinned staging buffer (record byte b lands at b - q)
    const uint64_t rec_bytes = total - q;
    BGP_TRY(c->h_stage.reserve_keep(c->h_stage_len + rec_bytes + 16, c->h_stage_len));
    uint8_t* dst = c->h_stage.as<uint8_t>() + c->h_stage_len;
    const auto t2 = now();
    if (head.size() > q) memcpy(dst, head.data() + q, head.size() - (size_t)q);       // tail of the last header block
    if (n_threads <= 0) n_threads = bgp_default_threads();
    if (n_threads < 1) n_threads = 1;
    std::atomic<size_t> next{jh}; std::atomic<int> status{BGP_OK};
    // Workers take BATCH consecutive blocks and pread their compressed bytes into a private buffer: touching the
    // file through the mapping costs a page fault per 4 KB, and those serialise across threads (measured: the
    // same 0.5 s on 16 and on 256 threads for 1.6 GB); a read() into memory that is already mapped does not.
    constexpr size_t BATCH = 8;
    auto worker = [&]() {
        std::vector<uint8_t> buf(BATCH * 65536 + 4096);
        for (;;) {
            const size_t j0 = next.fetch_add(BATCH);
            if (j0 >= blocks.size() || status.load() < 0) return;
            const size_t j1 = std::min(blocks.size(), j0 + BATCH);
            const uint64_t beg = blocks[j0].coff, end = blocks[j1 - 1].coff + blocks[j1 - 1].cdata_off + blocks[j1 - 1].cdata_len + 8;
            bool local = false;                                                            // fallback: the mapping
            if (end - beg <= buf.size()) {
                size_t got = 0;
                while (got < end - beg) { const ssize_t r = pread(fd, buf.data() + got, end - beg - got, (off_t)(beg + got)); if (r <= 0) break; got += (size_t)r; }
                local = got == end - beg;
            }
            for (size_t j = j0; j < j1; ++j) {
                const Blk& b = blocks[j];
                const uint8_t* cdata = local ? buf.data() + (b.coff - beg) + b.cdata_off : file + b.coff + b.cdata_off;
                const int e = inflate_cdata(cdata, b, dst + (b.uoff - q));
                if (e < 0) status.store(e);
            }
        }
    };
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>((size_t)n_threads, blocks.size() - jh + 1);
    for (int t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (verbose) fprintf(stderr, "[biodgpu] load_bam: %zu blocks, %.1f MB -> %.1f MB; block scan %.1f ms, header + pinned staging %.1f ms, inflate on %d threads %.1f ms\n",
                         blocks.size(), fsz / 1e6, total / 1e6, ms(t0, t1), ms(t1, t2), nt, ms(t2, now()));
    if (status.load() < 0) { c->last_error = "zlib: corrupted BGZF block"; return status.load(); }
    for (const Blk& b : blocks) {                         // where every staged byte came from (withOffsets, readrange.d:64-86)
        if (b.uoff + b.isize <= q) continue;
        const uint32_t skip = b.uoff < q ? (uint32_t)(q - b.uoff) : 0u;
        c->stage_map.push_back(bgp_ctx::StageSeg{c->h_stage_len + (b.uoff + skip - q), b.isize - skip, b.coff, skip});
    }
    c->bai_buildable = c->h_stage_len == 0;               // virtual offsets of ONE file: what createIndex needs
    c->stage_end_coff = 0;
    for (const Blk& b : blocks) if (b.isize) c->stage_end_coff = b.coff + b.cdata_off + b.cdata_len + 8;   // inputstream.d:447-450
    c->h_stage_len += rec_bytes;
    return BGP_OK;
}

// ---- device inflate: the file goes to HBM compressed, k_inflate (inflate.hip) does the rest -------------------------
//
// Phase A: reader threads pread 16 MB pieces into a pinned ring and push them to the device; the block headers are
// walked on the host as the pieces arrive, in file order (each header names the next), from the ring itself.
// Phase B: the BAM header is inflated on the host (its blocks only) to find the first record; every block from there
// on gets a descriptor (CDATA offset, output offset = uncompressed offset - first record, ISIZE, CRC) and one wave.
// No inflated byte exists on the host; the virtual-offset map is filled as in bgzf_load_bam.
#include <condition_variable>
#include <mutex>
#include <memory>

int bgzf_load_bam_device(bgp_ctx* c, const char* path) {
    const bool verbose = getenv("BGP_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { c->last_error = std::string("cannot open ") + path; return BGP_EINVAL; }
    struct CloseFd { int fd; ~CloseFd() { close(fd); } } close_fd{fd};
    struct stat sb;
    if (fstat(fd, &sb) != 0) { c->last_error = "fstat failed"; return BGP_EINVAL; }
    const uint64_t fsz = (uint64_t)sb.st_size;
    constexpr uint64_t CH = 16ull << 20, OV = 192ull << 10;
    const uint64_t n_chunks = (fsz + CH - 1) / CH;
    // readers: 4 and 8 threads both take ~65 ms for 1.6 GB on the measured box, 16 take 95, 24 take 140 (scratch/read_threads.sh)
    const uint64_t t_max = getenv("BGP_READ_THREADS") ? (uint64_t)atoi(getenv("BGP_READ_THREADS")) : 8ull;
    const int T = (int)std::max<uint64_t>(1, std::min<uint64_t>({t_max, (uint64_t)bgp_default_threads(), n_chunks}));
    BGP_TRY(c->d_comp.reserve(fsz + 1024));
    BGP_TRY(c->h_ring.reserve((size_t)T * (CH + OV)));
    const auto t1 = now();

    std::vector<Blk> blocks;
    std::mutex mu; std::condition_variable cv;
    uint64_t turn = 0, scan_off = 0, scan_uoff = 0; int scan_rc = BGP_OK; std::string scan_err;
    std::atomic<uint64_t> next{0}; std::atomic<int> hip_rc{BGP_OK};
    // The blocks are inflated WHILE the file is still arriving: this thread follows the scan, and whenever a few thousand blocks lie whole in
    // what has been uploaded it launches k_inflate for them on the ctx's stream, behind the uploads' events (read + upload and inflate took
    // turns before: 66 + 69 ms for a 1.6 GB file).  The output buffer and the descriptor table are sized by estimate -- the first chunk's
    // ratio and a third more, a block per 2 KB -- and a file that outgrows either is inflated the old way, in one launch at the end.
    std::vector<hipEvent_t> ev(n_chunks, nullptr);
    std::unique_ptr<std::atomic<int>[]> ev_ready(new std::atomic<int>[n_chunks]);
    struct Events { std::vector<hipEvent_t>& e; ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); } } events_guard{ev};      // (in front of the loop that makes them: an early return inside it lets go of the ones made so far)
    for (uint64_t k = 0; k < n_chunks; ++k) { ev_ready[k].store(0); if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { ev[k] = nullptr; c->last_error = "hipEventCreate failed"; return BGP_EDEVICE; } }
    // ONE stream for all uploads and two for the inflate, the ctx's own (bgp_init makes them, in front of any other stream: a stream per
    // reader and four for the batches were more streams than the device has hardware queues, and an upload that shared a queue with a
    // running k_inflate waited for it to end -- the uploads stalled for the 50 ms of the first batch; creating a dozen streams also took
    // the load's first 25 ms).
    std::mutex up_mu;                                                      // (a copy and its event are enqueued together)
    auto reader_ev = [&](int t) {      // reader() with the upload's event made known
        if (hipSetDevice(c->device) != hipSuccess) { hip_rc.store(BGP_EDEVICE); }
        uint8_t* buf = c->h_ring.as<uint8_t>() + (size_t)t * (CH + OV);
        int64_t last = -1;                                                 // the chunk whose upload reads this buffer
        for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= n_chunks) break;
            if (last >= 0 && hipEventSynchronize(ev[last]) != hipSuccess) hip_rc.store(BGP_EDEVICE);      // the previous piece has left this buffer
            const uint64_t lo = k * CH, hi = std::min(fsz, lo + CH + OV);
            uint64_t got = 0;
            while (got < hi - lo) { const ssize_t r = pread(fd, buf + got, hi - lo - got, (off_t)(lo + got)); if (r <= 0) break; got += (uint64_t)r; }
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return turn == k; });
                if (scan_rc == BGP_OK) {
                    if (got < hi - lo) { scan_rc = BGP_ETRUNC; scan_err = "short read"; }
                    else if (scan_off < lo) { scan_rc = BGP_EINVAL; scan_err = "BGZF block larger than 64 KB"; }
                    else scan_rc = scan_blocks(buf - lo, hi, std::min(fsz, lo + CH), scan_off, scan_uoff, blocks, scan_err);
                }
                turn = k + 1;
            }
            cv.notify_all();
            if (hip_rc.load() == BGP_OK) {
                std::lock_guard<std::mutex> ul(up_mu);
                if (hipMemcpyAsync(c->d_comp.as<uint8_t>() + lo, buf, std::min(CH, fsz - lo), hipMemcpyHostToDevice, c->io_up) != hipSuccess || hipEventRecord(ev[k], c->io_up) != hipSuccess) hip_rc.store(BGP_EDEVICE);
                else last = (int64_t)k;
            }
            ev_ready[k].store(1, std::memory_order_release);
        }
        if (last >= 0) (void)hipEventSynchronize(ev[last]);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(reader_ev, t);

    const bool pipelined = !(getenv("BGP_INFLATE_PIPELINE") && atoi(getenv("BGP_INFLATE_PIPELINE")) == 0);
    constexpr uint64_t SLACK = 65536;                     // the block that holds the first record starts up to 64 KB before it
    uint64_t est_bytes = 0;                                // the inflated size, estimated from the blocks scanned when the header has parsed (their ratio and a third more)
    const size_t nb_cap = (size_t)(fsz / 2048 + 4096);
    std::vector<bgp_inflate_block> desc;
    std::vector<uint64_t> desc_src_blk;                    // (for error messages: nothing else)
    desc.reserve(nb_cap);
    int32_t* d_status = nullptr;
    bool fallback = !pipelined, have_q = false;
    uint64_t q = 0;
    size_t bi = 0, n_desc_launched = 0;                    // blocks looked at; descriptors already launched
    uint64_t up_chunks = 0;
    std::vector<uint8_t> head, cbuf(65536 + 64);
    size_t jh = 0;
    int disp_rc = BGP_OK;
    std::vector<Blk> snap;
    // (a batch goes to whichever of the two inflate streams has run dry and takes every block that is there: the upload outpaces the
    // inflate, so the batches grow, and while one drains -- a block is 9-12 ms of one wave's work whatever runs beside it -- the other
    // stream's fills the wave slots it leaves.  Batches down ONE stream left the device half empty for a block's time at every seam.)
    const size_t BATCH_MIN = getenv("BGP_INFLATE_BATCH_MIN") ? (size_t)std::max(1, atoi(getenv("BGP_INFLATE_BATCH_MIN"))) : 2048;      // (tests: many small batches from a small file)
    int n_batch = 0;
    for (;;) {
        // (the reader threads are joinable from here on: an exception out of this loop -- its vectors grow -- would unwind past live
        // std::thread objects and end the process; it becomes a status, the readers are told to stop, and the join below still happens)
        try {
        uint64_t tn; int src; size_t nblk;
        {
            std::unique_lock<std::mutex> lk(mu);
            tn = turn; src = scan_rc; nblk = blocks.size();
            if (nblk > snap.size()) snap.insert(snap.end(), blocks.begin() + (ptrdiff_t)snap.size(), blocks.end());
        }
        const bool all_scanned = tn >= n_chunks || src != BGP_OK || hip_rc.load() != BGP_OK;
        if (src != BGP_OK || hip_rc.load() != BGP_OK) break;
        if (!fallback && disp_rc == BGP_OK) {
            // the BAM header: inflate its blocks on the host until it parses (reader.d:111-116,579-597)
            while (!have_q && jh < snap.size()) {
                const Blk& b = snap[jh];
                head.resize((size_t)(b.uoff + b.isize));
                if ((uint64_t)pread(fd, cbuf.data(), b.cdata_len, (off_t)(b.coff + b.cdata_off)) != b.cdata_len) { c->last_error = "short read"; disp_rc = BGP_ETRUNC; break; }
                const int rc = inflate_cdata(cbuf.data(), b, head.data() + b.uoff);
                if (rc < 0) { c->last_error = "zlib: corrupted BGZF block"; disp_rc = rc; break; }
                ++jh;
                const int got = bgp_parse_bam_header(head.data(), head.size(), c->refs, &q, c->last_error);
                if (got < 0) { disp_rc = got; break; }
                if (got > 0) {
                    have_q = true;
                    if (verbose) fprintf(stderr, "[biodgpu]   header parsed at %.2f ms\n", ms(t1, now()));
                    {   // (a BAM compresses about evenly: what the first chunk inflates to says what the file will; a file that outgrows it is inflated at the end)
                        const Blk& lb = snap.back();
                        const double ratio = (double)(lb.uoff + lb.isize) / (double)std::max<uint64_t>(1, lb.coff + lb.cdata_off + lb.cdata_len + 8);
                        est_bytes = (uint64_t)((double)fsz * ratio * 1.34) + (64ull << 20);
                    }
                    if (c->d_stream_own.reserve(SLACK + est_bytes + 256) != BGP_OK || c->d_inf_blk.reserve(nb_cap * (sizeof(bgp_inflate_block) + 4) + 64) != BGP_OK) { fallback = true; (void)hipGetLastError(); }      // (the estimate did not fit: the failed hipMalloc's error must not be what the next launch check reads)
                    else d_status = (int32_t*)(c->d_inf_blk.as<uint8_t>() + nb_cap * sizeof(bgp_inflate_block));
                }
            }
            while (up_chunks < tn && ev_ready[up_chunks].load(std::memory_order_acquire)) ++up_chunks;
            const uint64_t up_bytes = std::min(fsz, up_chunks * CH);
            if (have_q && !fallback && disp_rc == BGP_OK) {
                size_t j = bi; const size_t d0 = desc.size();
                uint64_t bytes = 0;
                while (j < snap.size() && snap[j].coff + snap[j].cdata_off + snap[j].cdata_len + 8 <= up_bytes) {
                    const Blk& b = snap[j];
                    if (b.isize && b.uoff + b.isize > q) {
                        if (desc.size() >= nb_cap || b.uoff + b.isize - q > est_bytes) { fallback = true; break; }
                        desc.push_back(bgp_inflate_block{b.coff + b.cdata_off, SLACK + b.uoff - q, b.cdata_len, b.isize, b.crc, 0});
                        bytes += b.isize;
                    }
                    ++j;
                }
                const size_t nd = desc.size() - d0;
                hipStream_t is = nullptr;
                if (!fallback && nd) {
                    const bool last = all_scanned && j == snap.size();
                    for (hipStream_t x : c->io_inf) if (!is && hipStreamQuery(x) == hipSuccess) is = x;
                    if (!is && last) is = c->io_inf[n_batch & 1];
                    if (is && !last && nd < BATCH_MIN) is = nullptr;
                }
                if (is) {
                    const uint64_t k0 = snap[bi].coff / CH;
                    bool ok = true;
                    ++n_batch;
                    if (verbose) fprintf(stderr, "[biodgpu]   batch %d at %.2f ms: %zu blocks, %llu chunks uploaded, scanned %llu\n", n_batch - 1, ms(t1, now()), nd, (unsigned long long)up_chunks, (unsigned long long)tn);
                    for (uint64_t k = k0; k < up_chunks && ok; ++k) ok = hipStreamWaitEvent(is, ev[k], 0) == hipSuccess;
                    ok = ok && hipMemcpyAsync(c->d_inf_blk.as<bgp_inflate_block>() + d0, desc.data() + d0, nd * sizeof(bgp_inflate_block), hipMemcpyHostToDevice, is) == hipSuccess;
                    if (!ok) { c->last_error = "HIP error while launching the inflate"; disp_rc = BGP_EDEVICE; }
                    else { const int rc = inflate_blocks_device(c, c->d_comp.as<uint8_t>(), c->d_stream_own.as<uint8_t>(), c->d_inf_blk.as<bgp_inflate_block>() + d0, (uint32_t)nd, d_status + d0, bytes, is); if (rc != BGP_OK) disp_rc = rc; }
                    bi = j; n_desc_launched = desc.size();
                } else if (!fallback && nd == 0) bi = j;    // (blocks of the header, empty blocks: nothing to launch for them)
                else if (!fallback) desc.resize(d0);        // (not enough yet: looked at again next time round)
            }
        }
        if (all_scanned && (fallback || disp_rc != BGP_OK || (have_q && bi == snap.size() && up_chunks >= n_chunks))) break;
        if (all_scanned && !have_q) break;                 // (no header in the whole file: reported below)
        std::this_thread::sleep_for(std::chrono::microseconds(150));
        } catch (const std::bad_alloc&) { disp_rc = BGP_ENOMEM; try { c->last_error = "out of host memory while loading the file"; } catch (...) { } next.store(n_chunks); break; }
          catch (...) { disp_rc = BGP_EINVAL; try { c->last_error = "unexpected C++ exception while loading the file"; } catch (...) { } next.store(n_chunks); break; }
    }
    for (auto& x : th) x.join();
    (void)hipStreamSynchronize(c->io_up); for (hipStream_t x : c->io_inf) (void)hipStreamSynchronize(x);
    if (hip_rc.load() < 0) { (void)hipStreamSynchronize(c->stream); c->last_error = "HIP error while uploading the compressed file"; return hip_rc.load(); }
    if (scan_rc < 0) { (void)hipStreamSynchronize(c->stream); c->last_error = scan_err; return scan_rc; }
    if (disp_rc < 0) { (void)hipStreamSynchronize(c->stream); return disp_rc; }
    const auto t2 = now();

    const uint64_t total = blocks.empty() ? 0 : blocks.back().uoff + blocks.back().isize;
    if (!have_q || fallback) {
        // everything at the end, in one launch (the header did not parse on the way, or the file outgrew the estimates)
        BGP_HIP(hipStreamSynchronize(c->stream));
        head.clear(); q = 0; jh = 0; int got = 0;
        c->refs.clear();
        while (jh < blocks.size() && got == 0) {
            const Blk& b = blocks[jh];
            head.resize((size_t)(b.uoff + b.isize));
            if ((uint64_t)pread(fd, cbuf.data(), b.cdata_len, (off_t)(b.coff + b.cdata_off)) != b.cdata_len) { c->last_error = "short read"; return BGP_ETRUNC; }
            const int rc = inflate_cdata(cbuf.data(), b, head.data() + b.uoff);
            if (rc < 0) { c->last_error = "zlib: corrupted BGZF block"; return rc; }
            ++jh;
            got = bgp_parse_bam_header(head.data(), head.size(), c->refs, &q, c->last_error);
            if (got < 0) return got;
        }
        if (got == 0) { c->last_error = total < 4 ? "wrong BAM magic" : "truncated BAM header"; return total < 4 ? BGP_EINVAL : BGP_ETRUNC; }
    }
    c->n_ref_hint = (int32_t)c->refs.size();
    const uint64_t rec_bytes = total - q;
    std::vector<int32_t> st;
    if (fallback || !have_q) {
        BGP_TRY(c->d_stream_own.reserve(SLACK + rec_bytes + 256));
        desc.clear();
        for (const Blk& b : blocks) {
            if (!b.isize || b.uoff + b.isize <= q) continue;
            desc.push_back(bgp_inflate_block{b.coff + b.cdata_off, SLACK + b.uoff - q, b.cdata_len, b.isize, b.crc, 0});
        }
        const uint32_t nb = (uint32_t)desc.size();
        BGP_TRY(c->d_inf_blk.reserve((size_t)nb * (sizeof(bgp_inflate_block) + 4) + 64));
        d_status = (int32_t*)(c->d_inf_blk.as<uint8_t>() + (size_t)nb * sizeof(bgp_inflate_block));
        st.assign(nb, 0);
        if (nb) {
            BGP_HIP(hipMemcpyAsync(c->d_inf_blk.p, desc.data(), (size_t)nb * sizeof(bgp_inflate_block), hipMemcpyHostToDevice, c->stream));
            BGP_TRY(inflate_blocks_device(c, c->d_comp.as<uint8_t>(), c->d_stream_own.as<uint8_t>(), c->d_inf_blk.as<bgp_inflate_block>(), nb, d_status, rec_bytes));
            BGP_HIP(hipMemcpyAsync(st.data(), d_status, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        }
    } else {
        st.assign(desc.size(), 0);
        if (!desc.empty()) BGP_HIP(hipMemcpyAsync(st.data(), d_status, desc.size() * 4, hipMemcpyDeviceToHost, c->stream));
        (void)n_desc_launched;
    }
    BGP_HIP(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < st.size(); ++i)
        if (st[i]) { c->last_error = "zlib: corrupted BGZF block (device inflate, block at " + std::to_string(desc[i].src) + ", reason " + std::to_string(st[i]) + ")"; return BGP_EINVAL; }
    if (verbose) fprintf(stderr, "[biodgpu] load_bam_device: %zu blocks, %.1f MB -> %.1f MB; buffers %.1f ms, read + scan + upload on %d threads %.1f ms, header + inflate %.1f ms\n",
                         blocks.size(), fsz / 1e6, total / 1e6, ms(t0, t1), T, ms(t1, t2), ms(t2, now()));
    for (const Blk& b : blocks) {                         // where every stream byte came from (withOffsets, readrange.d:64-86)
        if (b.uoff + b.isize <= q) continue;
        const uint32_t skip = b.uoff < q ? (uint32_t)(q - b.uoff) : 0u;
        c->stage_map.push_back(bgp_ctx::StageSeg{b.uoff + skip - q, b.isize - skip, b.coff, skip});
    }
    c->bai_buildable = true; c->stage_end_coff = 0;
    for (const Blk& b : blocks) if (b.isize) c->stage_end_coff = b.coff + b.cdata_off + b.cdata_len + 8;
    c->d_stream = c->d_stream_own.as<uint8_t>() + SLACK; c->stream_len = rec_bytes; c->stream_on_device = true;
    return BGP_OK;
}

// BAM magic, header text and reference table (reader.d:111-116,579-597) of the file behind fd, its first blocks inflated on the
// host until they parse: *q = offset of the first record in the inflated stream.
int bgzf_read_bam_header(int fd, uint64_t fsz, std::vector<RefInfo>& refs, uint64_t* q, std::string& err) {
    std::vector<uint8_t> raw, head;
    std::vector<Blk> blocks;
    uint64_t have = 0, off = 0, uoff = 0; size_t jh = 0;
    for (uint64_t want = 1ull << 20;; want *= 4) {
        const uint64_t upto = std::min(fsz, want);
        raw.resize((size_t)upto);
        while (have < upto) { const ssize_t r = pread(fd, raw.data() + have, (size_t)(upto - have), (off_t)have); if (r <= 0) { err = "short read"; return BGP_ETRUNC; } have += (uint64_t)r; }
        // (blocks that lie whole in what has been read: a header cut by `upto` is a reason to read more, not an error, unless the file ends there)
        const int rc = scan_blocks(raw.data(), upto, upto < fsz ? (upto > 65536 + 26 ? upto - 65536 - 26 : 0) : upto, off, uoff, blocks, err);
        if (rc < 0) return rc;
        while (jh < blocks.size()) {
            const Blk& b = blocks[jh];
            head.resize((size_t)(b.uoff + b.isize));
            if (inflate_cdata(raw.data() + b.coff + b.cdata_off, b, head.data() + b.uoff) < 0) { err = "zlib: corrupted BGZF block"; return BGP_EINVAL; }
            ++jh;
            const int got = bgp_parse_bam_header(head.data(), head.size(), refs, q, err);
            if (got < 0) return got;
            if (got > 0) return BGP_OK;
        }
        if (upto == fsz) break;
    }
    const uint64_t total = blocks.empty() ? 0 : blocks.back().uoff + blocks.back().isize;
    err = total < 4 ? "wrong BAM magic" : "truncated BAM header";
    return total < 4 ? BGP_EINVAL : BGP_ETRUNC;
}
