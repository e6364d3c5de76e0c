// bai_host.cpp -- BAI index, region -> chunk list, chunk list -> inflated record bytes.
//
// Host-side mirror of what BioD runs above the hot path for bam["chr"][beg .. end]:
//   BaiFile           bio/std/hts/bam/baifile.d:92-169        (file name search, parse of section 4.2 of the spec)
//   Index.getMinimumOffset                        :77-82       (linear index lookup)
//   Bin.canOverlapWith / reg2bin  bio/std/hts/bam/bai/bin.d:55-92
//   RandomAccessManager.getChunks bio/std/hts/bam/randomaccessmanager.d:209-244 (bins -> chunks, clamp, sort, merge)
//   nonOverlappingChunks          bio/core/utils/algo.d:95-168
//   StreamChunksSupplier          bio/core/bgzf/inputstream.d:254-330 (which bytes of which blocks a chunk list yields)
// The records are then framed, filtered (region.hip) and piled up on the device like any other staged stream.
#include "common.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <memory>
#include <thread>

struct BaiBin { uint32_t id; std::vector<bgp_chunk> chunks; };
struct BaiIndex { std::vector<BaiBin> bins; std::vector<uint64_t> ioffsets; };
struct bgp_bai { std::vector<BaiIndex> indices; std::vector<bgp_chunk> out; std::string error; };

namespace {

constexpr uint32_t BAI_MAX_BIN_ID = 37449;           // bio/std/hts/bam/constants.d:35-37
constexpr int BAI_LINEAR_INDEX_WINDOW_SIZE = 16384;

struct Cursor {
    const uint8_t* p; size_t n, off = 0; bool ok = true;
    template <class T> T get() { T v{}; if (off + sizeof(T) > n) { ok = false; return v; } memcpy(&v, p + off, sizeof(T)); off += sizeof(T); return v; }
};

bool file_exists(const std::string& f) { struct stat sb; return stat(f.c_str(), &sb) == 0 && S_ISREG(sb.st_mode); }

bool read_file(const std::string& f, std::vector<uint8_t>& out) {
    FILE* fp = fopen(f.c_str(), "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END); const long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
    out.resize(sz > 0 ? (size_t)sz : 0);
    const bool ok = out.empty() || fread(out.data(), 1, out.size(), fp) == out.size();
    fclose(fp);
    return ok;
}

bool chunk_less(const bgp_chunk& a, const bgp_chunk& b) { return a.beg != b.beg ? a.beg < b.beg : a.end < b.end; }   // Chunk.opCmp, bio/core/bgzf/chunk.d:33-39

}  // namespace

extern "C" {

int bgp_reg2bin(int32_t beg, int32_t end) {                    // bai/bin.d:82-92
    BGP_API_TRY
    if (end == beg) end = beg + 1;
    --end;
    if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

int bgp_bin_can_overlap(uint32_t id, int32_t begin, int32_t end) {   // Bin.canOverlapWith, bai/bin.d:55-77
    BGP_API_TRY
    if (id == 0) return 1;
    if (id > BAI_MAX_BIN_ID) return 0;
    if (begin < 0) begin = 0;
    uint32_t magic = 4681;
    int32_t b = begin >> 14, e = end >> 14;
    for (;;) {
        // `id - magic_number` is uint arithmetic in D and the comparisons with the int bounds are unsigned
        const uint32_t delta = id - magic;
        if ((uint32_t)b <= delta && delta <= (uint32_t)e) return 1;
        magic >>= 3;
        if (magic == 0) return 0;
        b >>= 3; e >>= 3;
    }
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

int bgp_bai_open(const char* path, bgp_bai** out) {            // BaiFile(filename), baifile.d:92-169
    BGP_API_TRY
    if (!path || !out) return BGP_EINVAL;
    *out = nullptr;
    std::string fn(path), use;
    const bool is_bai = fn.size() >= 4 && fn.compare(fn.size() - 4, 4, ".bai") == 0;
    if (is_bai) use = fn;
    else {
        const std::string first = fn + ".bai";
        const size_t dot = fn.rfind('.');
        const std::string second = (dot == std::string::npos ? std::string() : fn.substr(0, dot + 1)) + "bai";   // retro(find(retro(f), '.')) ~ "bai"
        if (file_exists(first)) use = first;
        else if (file_exists(second)) use = second;
        else return BGP_EINVAL;
    }
    std::vector<uint8_t> buf;
    if (!read_file(use, buf)) return BGP_EINVAL;
    Cursor c{buf.data(), buf.size()};
    if (buf.size() < 8 || memcmp(buf.data(), "BAI\1", 4) != 0) return BGP_EINVAL;
    c.off = 4;
    const int32_t n_ref = c.get<int32_t>();
    if (!c.ok || n_ref < 0 || (size_t)n_ref > (buf.size() - 8) / 8) return BGP_EINVAL;     // every reference takes at least n_bin + n_intv = 8 bytes
    std::unique_ptr<bgp_bai> bp(new bgp_bai());       // (an allocation failure further down unwinds through the ABI guard)
    bgp_bai* b = bp.get();
    b->indices.resize((size_t)n_ref);
    for (int32_t i = 0; i < n_ref && c.ok; ++i) {
        const int32_t n_bin = c.get<int32_t>();
        if (!c.ok || n_bin < 0 || (size_t)n_bin > buf.size()) { c.ok = false; break; }
        b->indices[i].bins.resize((size_t)n_bin);
        for (int32_t j = 0; j < n_bin && c.ok; ++j) {
            BaiBin& bin = b->indices[i].bins[j];
            bin.id = c.get<uint32_t>();
            const int32_t n_chunk = c.get<int32_t>();
            if (!c.ok || n_chunk < 0 || (size_t)n_chunk > buf.size()) { c.ok = false; break; }
            bin.chunks.resize((size_t)n_chunk);
            for (int32_t k = 0; k < n_chunk && c.ok; ++k) { bin.chunks[k].beg = c.get<uint64_t>(); bin.chunks[k].end = c.get<uint64_t>(); }
        }
        const int32_t n_intv = c.get<int32_t>();
        if (!c.ok || n_intv < 0 || (size_t)n_intv > buf.size()) { c.ok = false; break; }
        b->indices[i].ioffsets.resize((size_t)n_intv);
        for (int32_t j = 0; j < n_intv && c.ok; ++j) b->indices[i].ioffsets[j] = c.get<uint64_t>();
    }
    if (!c.ok) return BGP_ETRUNC;
    *out = bp.release();
    return BGP_OK;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

void bgp_bai_close(bgp_bai* b) { delete b; }

int bgp_bai_n_ref(const bgp_bai* b, int32_t* n) { if (!b || !n) return BGP_EINVAL; *n = (int32_t)b->indices.size(); return BGP_OK; }

int bgp_bai_shape(const bgp_bai* b, int32_t ref_id, uint32_t* n_bins, uint32_t* n_chunks, uint32_t* n_intervals) {
    BGP_API_TRY
    if (!b || ref_id < 0 || (size_t)ref_id >= b->indices.size()) return BGP_EINVAL;
    const BaiIndex& ix = b->indices[ref_id];
    uint32_t nc = 0;
    for (const BaiBin& bin : ix.bins) nc += (uint32_t)bin.chunks.size();
    if (n_bins) *n_bins = (uint32_t)ix.bins.size();
    if (n_chunks) *n_chunks = nc;
    if (n_intervals) *n_intervals = (uint32_t)ix.ioffsets.size();
    return BGP_OK;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

int bgp_bai_min_offset(const bgp_bai* b, int32_t ref_id, int32_t position, uint64_t* voffset) {   // baifile.d:77-82
    BGP_API_TRY
    if (!b || !voffset || ref_id < 0 || (size_t)ref_id >= b->indices.size()) return BGP_EINVAL;
    const BaiIndex& ix = b->indices[ref_id];
    const int pos = std::max(0, position);
    const int i = std::min(pos / BAI_LINEAR_INDEX_WINDOW_SIZE, (int)ix.ioffsets.size() - 1);
    *voffset = i == -1 ? 0 : ix.ioffsets[(size_t)i];
    return BGP_OK;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

int bgp_bai_chunks(bgp_bai* b, int32_t ref_id, uint32_t beg, uint32_t end, const bgp_chunk** chunks, uint32_t* n) {   // getChunks, randomaccessmanager.d:222-244
    BGP_API_TRY
    if (!b || !chunks || !n) return BGP_EINVAL;
    if (ref_id < 0 || (size_t)ref_id >= b->indices.size()) return BGP_EINVAL;          // checkRefId, :205-207
    uint64_t min_offset = 0;
    bgp_bai_min_offset(b, ref_id, (int32_t)beg, &min_offset);
    std::vector<bgp_chunk> v;
    for (const BaiBin& bin : b->indices[ref_id].bins) {
        if (!bgp_bin_can_overlap(bin.id, (int32_t)beg, (int32_t)end)) continue;
        for (const bgp_chunk& ch : bin.chunks)                                          // appendChunks, :209-219
            if (ch.end > min_offset) { v.push_back(ch); if (v.back().beg < min_offset) v.back().beg = min_offset; }
    }
    std::sort(v.begin(), v.end(), chunk_less);
    b->out.clear();
    for (size_t i = 0; i < v.size();) {                                                 // nonOverlappingChunks, algo.d:95-147
        bgp_chunk f = v[i++];
        while (i < v.size() && f.end >= v[i].beg) { f.end = std::max(f.end, v[i].end); ++i; }
        b->out.push_back(f);
    }
    *chunks = b->out.data(); *n = (uint32_t)b->out.size();
    return BGP_OK;
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ chunk list -> bytes
namespace {

struct Piece { uint64_t coff; uint32_t cdata_off, cdata_len, crc, isize, skip, keep; uint64_t dst; };

// One BGZF block header at file offset `off` (inputstream.d:54-199); returns the offset of the next block.
int block_at(const uint8_t* f, uint64_t n, uint64_t off, Piece& b, uint64_t* next, std::string& err) {
    if (n - off < 12 || !(f[off] == 0x1F && f[off + 1] == 0x8B && f[off + 2] == 8 && f[off + 3] == 4)) { err = "wrong BGZF magic"; return BGP_EINVAL; }
    const uint32_t xlen = f[off + 10] | (f[off + 11] << 8);
    uint64_t p = off + 12; const uint64_t xend = p + xlen;
    if (xend > n) { err = "truncated BGZF extra field"; return BGP_ETRUNC; }
    int bsize = -1; uint32_t len = 0;
    while (len < xlen) {
        if (p + 4 > n) { err = "truncated BGZF subfield"; return BGP_ETRUNC; }
        const uint32_t slen = f[p + 2] | (f[p + 3] << 8);
        if (f[p] == 66 && f[p + 1] == 67) {
            if (slen != 2) { err = "wrong BC subfield length"; return BGP_EINVAL; }
            if (bsize >= 0) { err = "duplicate field with block size"; return BGP_EINVAL; }
            if (p + 6 > n) { err = "truncated BC subfield"; return BGP_ETRUNC; }
            bsize = f[p + 4] | (f[p + 5] << 8);
        }
        p += 4 + slen; len += 4 + slen;
    }
    if (len != xlen || bsize < 0) { err = "bad BGZF extra field"; return BGP_EINVAL; }
    const int64_t cdata = (int64_t)bsize - xlen - 19;
    if (cdata < 0 || cdata > 65536) { err = "bad compressed data size"; return BGP_EINVAL; }
    if (xend + (uint64_t)cdata + 8 > n) { err = "truncated BGZF block"; return BGP_ETRUNC; }
    b.coff = off; b.cdata_off = (uint32_t)(xend - off); b.cdata_len = (uint32_t)cdata;
    memcpy(&b.crc, f + xend + cdata, 4); memcpy(&b.isize, f + xend + cdata + 4, 4);
    if (b.isize > 65536) { err = "uncompressed block size must be within 65536 bytes"; return BGP_EINVAL; }
    *next = xend + (uint64_t)cdata + 8;
    return BGP_OK;
}

struct MappedFile {
    const uint8_t* p = nullptr; size_t n = 0;
    int open(const char* path, std::string& err) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) { err = std::string("cannot open ") + path; return BGP_EINVAL; }
        struct stat sb;
        if (fstat(fd, &sb) != 0) { close(fd); err = "fstat failed"; return BGP_EINVAL; }
        n = (size_t)sb.st_size;
        p = n ? (const uint8_t*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
        close(fd);
        if (n && p == (const uint8_t*)MAP_FAILED) { p = nullptr; err = "mmap failed"; return BGP_EINVAL; }
        return BGP_OK;
    }
    ~MappedFile() { if (p && n) munmap((void*)p, n); }
};

// Which slice of which block every chunk contributes (StreamChunksSupplier.getNextBgzfBlock, inputstream.d:281-324):
// the first block of a chunk loses beg.uoffset leading bytes, the block at end.coffset keeps end.uoffset bytes;
// leading chunks that start in the same block are folded into the last of them (moveToNextChunk, :259-273).
int plan_chunks(const MappedFile& mf, const bgp_chunk* chunks, uint32_t n, std::vector<Piece>& pieces, uint64_t* total, std::string& err) {
    std::vector<bgp_chunk> q(chunks, chunks + n);
    uint64_t dst = 0;
    size_t at = 0;
    auto move_to_next = [&]() {
        if (at >= q.size()) return;
        const uint64_t beg = q[at].beg;
        size_t i = at + 1;
        for (; i < q.size(); ++i) if ((q[i].beg >> 16) > (q[at].beg >> 16)) break;
        at = i - 1; q[at].beg = beg;
    };
    move_to_next();
    while (at < q.size()) {
        const bgp_chunk ch = q[at];
        uint64_t off = ch.beg >> 16;
        for (;;) {
            if (off >= mf.n) break;                                   // fillBgzfBufferFromStream: no more blocks -> supplier ends
            Piece b{}; uint64_t next = 0;
            const int rc = block_at(mf.p, mf.n, off, b, &next, err);
            if (rc < 0) return rc;
            const uint32_t skip_start = off == (ch.beg >> 16) ? (uint32_t)(ch.beg & 0xFFFF) : 0u;
            if (skip_start > b.isize || (off == (ch.end >> 16) && (uint32_t)(ch.end & 0xFFFF) > b.isize)) {     // an index that does not belong to this file
                err = "chunk offset past the end of its BGZF block";
                return BGP_EINVAL;
            }
            const int64_t skip_end = off == (ch.end >> 16) ? (int64_t)b.isize - (int64_t)(ch.end & 0xFFFF) : 0;
            const int64_t keep = (int64_t)b.isize - skip_start - skip_end;
            if (keep > 0) { b.skip = skip_start; b.keep = (uint32_t)keep; b.dst = dst; dst += (uint64_t)keep; pieces.push_back(b); }
            if (off >= (ch.end >> 16)) break;
            off = next;
        }
        ++at;
        move_to_next();
    }
    *total = dst;
    return BGP_OK;
}

int inflate_piece(const uint8_t* f, const Piece& b, uint8_t* dst, uint8_t* scratch) {      // block.d:127-216
    if (b.isize == 0) return BGP_OK;
    const bool direct = b.skip == 0 && b.keep == b.isize;
    uint8_t* out = direct ? dst : scratch;
    z_stream zs; memset(&zs, 0, sizeof zs);
    zs.next_in = const_cast<Bytef*>(f + b.coff + b.cdata_off); zs.avail_in = b.cdata_len;
    if (inflateInit2(&zs, -15) != Z_OK) return BGP_ENOMEM;
    zs.next_out = out; zs.avail_out = b.isize;
    const int e = inflate(&zs, Z_FINISH);
    const uLong produced = zs.total_out;
    inflateEnd(&zs);
    if (e != Z_STREAM_END || produced != b.isize) return BGP_EINVAL;
    if ((uint32_t)crc32(0, out, b.isize) != b.crc) return BGP_EINVAL;
    if (!direct) memcpy(dst, scratch + b.skip, b.keep);
    return BGP_OK;
}

int inflate_pieces(const MappedFile& mf, const std::vector<Piece>& pieces, uint8_t* dst, int n_threads) {
    if (n_threads <= 0) n_threads = bgp_default_threads();
    n_threads = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, pieces.size()));
    std::atomic<size_t> next{0}; std::atomic<int> status{BGP_OK};
    auto worker = [&]() {
        std::vector<uint8_t> scratch(65536);
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= pieces.size() || status.load() < 0) return;
            const int e = inflate_piece(mf.p, pieces[j], dst + pieces[j].dst, scratch.data());
            if (e < 0) status.store(e);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return status.load();
}

}  // namespace

extern "C" int bgp_bgzf_read_chunks(const char* path, const bgp_chunk* chunks, uint32_t n, int n_threads, uint8_t* out, uint64_t cap, uint64_t* len) {
    BGP_API_TRY
    if (!path || (!chunks && n) || !len) return BGP_EINVAL;
    MappedFile mf; std::string err;
    BGP_TRY(mf.open(path, err));
    std::vector<Piece> pieces; uint64_t total = 0;
    BGP_TRY(plan_chunks(mf, chunks, n, pieces, &total, err));
    *len = total;
    if (!out || cap < total) return BGP_OK;                 // size query
    return inflate_pieces(mf, pieces, out, n_threads);
    BGP_API_CATCH((bgp_ctx*)nullptr)
}

// Stage the records of a chunk list into the ctx (like bgzf_load_bam) and remember where every staged byte came
// from, so that BamReadBlock.start_virtual_offset (withOffsets, readrange.d:64-86) can be answered per record.
int bai_stage_chunks(bgp_ctx* c, const char* path, const bgp_chunk* chunks, uint32_t n, int n_threads) {
    MappedFile mf;
    BGP_TRY(mf.open(path, c->last_error));
    std::vector<Piece> pieces; uint64_t total = 0;
    BGP_TRY(plan_chunks(mf, chunks, n, pieces, &total, c->last_error));
    BGP_TRY(c->h_stage.reserve_keep(c->h_stage_len + total + 16, c->h_stage_len));
    const int rc = inflate_pieces(mf, pieces, c->h_stage.as<uint8_t>() + c->h_stage_len, n_threads);
    if (rc < 0) { c->last_error = "zlib: corrupted BGZF block"; return rc; }
    for (const Piece& p : pieces) c->stage_map.push_back(bgp_ctx::StageSeg{c->h_stage_len + p.dst, p.keep, p.coff, p.skip});
    c->h_stage_len += total;
    return BGP_OK;
}

// The same chunk list with decompressBgzfBlock on the DEVICE (bgp_load_bam_region_device): the chunks' blocks go to HBM compressed --
// reader threads pread the file ranges the chunks span, 8 MiB at a time, through pinned buffers --, k_inflate (inflate.hip: a wave a
// block, size and CRC32 checked) writes every whole block where its bytes belong in the record stream and the few cut ones (a chunk's
// first and last block: beg.uoffset bytes dropped, end.uoffset kept) into slots of their own, from which k_gather_pieces copies what the
// chunk keeps.  With eight ranks a node reading their shards of one file the host's cores are the ranks': no zlib runs on them.
namespace {
__global__ void k_gather_pieces(const uint8_t* slots, const uint4* cut, uint32_t n_cut, uint8_t* out) {      // cut[j] = (slot, skip, keep, dst lo) / dst hi in the next
    const uint4 a = cut[2 * blockIdx.x], b = cut[2 * blockIdx.x + 1];
    const uint8_t* src = slots + (uint64_t)a.x * 65536 + a.y;
    uint8_t* dst = out + (((uint64_t)b.x << 32) | a.w);
    for (uint32_t k = threadIdx.x; k < a.z; k += blockDim.x) dst[k] = src[k];
    (void)n_cut;
}
}  // namespace
int bai_stage_chunks_device(bgp_ctx* c, const char* path, const bgp_chunk* chunks, uint32_t n) {
    MappedFile mf;
    BGP_TRY(mf.open(path, c->last_error));
    std::vector<Piece> pieces; uint64_t total = 0;
    BGP_TRY(plan_chunks(mf, chunks, n, pieces, &total, c->last_error));
    if (pieces.empty()) { BGP_TRY(c->d_stream_own.reserve(65536 + 256)); c->d_stream = c->d_stream_own.as<uint8_t>(); c->stream_len = 0; c->stream_on_device = true; return BGP_OK; }      // (no chunk: no reads)
    // file ranges of consecutive blocks, each placed one behind the other in the device's copy
    struct Range { uint64_t lo, hi, at; };
    std::vector<Range> ranges; std::vector<uint64_t> comp_at(pieces.size());
    uint64_t comp_bytes = 0;
    for (size_t j = 0; j < pieces.size(); ++j) {
        const Piece& p = pieces[j];
        const uint64_t end = p.coff + p.cdata_off + p.cdata_len + 8;
        if (ranges.empty() || ranges.back().hi != p.coff) { ranges.push_back(Range{p.coff, end, comp_bytes}); }
        else ranges.back().hi = end;
        comp_at[j] = ranges.back().at + (p.coff - ranges.back().lo) + p.cdata_off;
        comp_bytes = ranges.back().at + (ranges.back().hi - ranges.back().lo);
    }
    constexpr uint64_t PART = 8ull << 20, SLACK = 65536;
    struct Part { uint64_t file_off, dev_off, len; };
    std::vector<Part> parts;
    for (const Range& r : ranges) for (uint64_t o = r.lo; o < r.hi; o += PART) parts.push_back(Part{o, r.at + (o - r.lo), std::min(PART, r.hi - o)});
    const int T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)bgp_default_threads(), parts.size()}));
    BGP_TRY(c->d_comp.reserve(comp_bytes + 1024));
    BGP_TRY(c->h_ring.reserve((size_t)T * PART));
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) { c->last_error = std::string("cannot open ") + path; return BGP_EINVAL; }
    struct CloseFd { int fd; ~CloseFd() { close(fd); } } close_fd{fd};
    std::atomic<size_t> next{0}; std::atomic<int> status{BGP_OK};
    std::mutex up_mu;
    auto reader = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) { status.store(BGP_EDEVICE); return; }
        uint8_t* buf = c->h_ring.as<uint8_t>() + (size_t)t * PART;
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { status.store(BGP_EDEVICE); return; }
        bool pending = false;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= parts.size() || status.load() < 0) break;
            if (pending && hipEventSynchronize(ev) != hipSuccess) { status.store(BGP_EDEVICE); break; }      // the previous part has left this buffer
            const Part& p = parts[k];
            uint64_t got = 0;
            while (got < p.len) { const ssize_t r = pread(fd, buf + got, p.len - got, (off_t)(p.file_off + got)); if (r <= 0) break; got += (uint64_t)r; }
            if (got < p.len) { status.store(BGP_ETRUNC); break; }
            std::lock_guard<std::mutex> ul(up_mu);
            if (hipMemcpyAsync(c->d_comp.as<uint8_t>() + p.dev_off, buf, p.len, hipMemcpyHostToDevice, c->io_up) != hipSuccess || hipEventRecord(ev, c->io_up) != hipSuccess) { status.store(BGP_EDEVICE); break; }
            pending = true;
        }
        if (pending) (void)hipEventSynchronize(ev);
        (void)hipEventDestroy(ev);
    };
    {
        std::vector<std::thread> th;
        try { for (int t = 1; t < T; ++t) th.emplace_back(reader, t); } catch (...) { }
        reader(0);
        for (auto& x : th) x.join();
    }
    BGP_HIP(hipStreamSynchronize(c->io_up));
    if (status.load() < 0) { c->last_error = status.load() == BGP_ETRUNC ? "short read" : "HIP error while uploading the chunks' blocks"; return status.load(); }
    // descriptors: a whole block inflates into place, a cut one into a 64 KiB slot behind the stream
    std::vector<bgp_inflate_block> desc; std::vector<uint4> cut;
    uint32_t n_slots = 0;
    for (size_t j = 0; j < pieces.size(); ++j) if (!(pieces[j].skip == 0 && pieces[j].keep == pieces[j].isize)) ++n_slots;
    const uint64_t slots_at = (SLACK + total + 255) & ~255ull;
    BGP_TRY(c->d_stream_own.reserve(slots_at + (uint64_t)n_slots * 65536 + 256));
    uint32_t slot = 0;
    for (size_t j = 0; j < pieces.size(); ++j) {
        const Piece& p = pieces[j];
        if (!p.isize) continue;
        if (p.skip == 0 && p.keep == p.isize) desc.push_back(bgp_inflate_block{comp_at[j], SLACK + p.dst, p.cdata_len, p.isize, p.crc, 0});
        else {
            desc.push_back(bgp_inflate_block{comp_at[j], slots_at + (uint64_t)slot * 65536, p.cdata_len, p.isize, p.crc, 0});
            const uint64_t d = SLACK + p.dst;
            cut.push_back(make_uint4(slot, p.skip, p.keep, (uint32_t)d)); cut.push_back(make_uint4((uint32_t)(d >> 32), 0u, 0u, 0u));
            ++slot;
        }
    }
    const uint32_t nb = (uint32_t)desc.size();
    BGP_TRY(c->d_inf_blk.reserve((size_t)nb * (sizeof(bgp_inflate_block) + 4) + cut.size() * 16 + 64));
    int32_t* d_status = (int32_t*)(c->d_inf_blk.as<uint8_t>() + (size_t)nb * sizeof(bgp_inflate_block));
    uint4* d_cut = (uint4*)(c->d_inf_blk.as<uint8_t>() + (((size_t)nb * (sizeof(bgp_inflate_block) + 4) + 15) & ~(size_t)15));
    std::vector<int32_t> st(nb, 0);
    if (nb) {
        BGP_HIP(hipMemcpyAsync(c->d_inf_blk.p, desc.data(), (size_t)nb * sizeof(bgp_inflate_block), hipMemcpyHostToDevice, c->stream));
        if (!cut.empty()) BGP_HIP(hipMemcpyAsync(d_cut, cut.data(), cut.size() * 16, hipMemcpyHostToDevice, c->stream));
        BGP_TRY(inflate_blocks_device(c, c->d_comp.as<uint8_t>(), c->d_stream_own.as<uint8_t>(), c->d_inf_blk.as<bgp_inflate_block>(), nb, d_status, total));
        if (!cut.empty()) { hipLaunchKernelGGL(k_gather_pieces, dim3((unsigned)(cut.size() / 2)), dim3(256), 0, c->stream, c->d_stream_own.as<uint8_t>() + slots_at, d_cut, (uint32_t)(cut.size() / 2), c->d_stream_own.as<uint8_t>()); BGP_HIP(hipGetLastError()); }
        BGP_HIP(hipMemcpyAsync(st.data(), d_status, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
    }
    BGP_HIP(hipStreamSynchronize(c->stream));
    for (uint32_t i = 0; i < nb; ++i) if (st[i]) { c->last_error = "zlib: corrupted BGZF block (device inflate, reason " + std::to_string(st[i]) + ")"; return BGP_EINVAL; }
    for (const Piece& p : pieces) c->stage_map.push_back(bgp_ctx::StageSeg{p.dst, p.keep, p.coff, p.skip});
    c->d_stream = c->d_stream_own.as<uint8_t>() + SLACK; c->stream_len = total; c->stream_on_device = true;
    return BGP_OK;
}

// BAM header (magic, text, reference table) only: the reference names/lengths a region query needs.
int bai_load_header(bgp_ctx* c, const char* path) {
    MappedFile mf;
    BGP_TRY(mf.open(path, c->last_error));
    std::vector<uint8_t> head; uint64_t off = 0;
    std::vector<uint8_t> scratch(65536);
    while (off < mf.n) {
        Piece b{}; uint64_t next = 0;
        BGP_TRY(block_at(mf.p, mf.n, off, b, &next, c->last_error));
        const size_t at = head.size();
        head.resize(at + b.isize);
        b.skip = 0; b.keep = b.isize;
        if (inflate_piece(mf.p, b, head.data() + at, scratch.data()) < 0) { c->last_error = "zlib: corrupted BGZF block"; return BGP_EINVAL; }
        uint64_t q = 0;
        const int got = bgp_parse_bam_header(head.data(), head.size(), c->refs, &q, c->last_error);
        if (got < 0) return got;
        if (got == 1) { c->n_ref_hint = (int32_t)c->refs.size(); return BGP_OK; }
        off = next;
    }
    c->last_error = "truncated BAM header";
    return BGP_ETRUNC;
}

int bai_record_voffset(bgp_ctx* c, uint64_t staged_off, uint64_t* voffset) {
    const auto& m = c->stage_map;
    size_t lo = 0, hi = m.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (m[mid].staged + m[mid].len <= staged_off) lo = mid + 1; else hi = mid; }
    if (lo >= m.size() || staged_off < m[lo].staged) return BGP_ESTATE;
    *voffset = (m[lo].coff << 16) | (uint64_t)(m[lo].skip + (staged_off - m[lo].staged));
    return BGP_OK;
}

// RandomAccessManager.getReadAt (randomaccessmanager.d:156-164): the one record that starts at `voffset`.
int bai_stage_read_at(bgp_ctx* c, const char* path, uint64_t voffset) {
    MappedFile mf;
    BGP_TRY(mf.open(path, c->last_error));
    std::vector<uint8_t> got, scratch(65536), blk(65536);
    std::vector<bgp_ctx::StageSeg> segs;
    uint64_t off = voffset >> 16, need = 4;
    uint32_t skip = (uint32_t)(voffset & 0xFFFF);
    while (got.size() < need) {
        if (off >= mf.n) { c->last_error = "record runs past the end of the file"; return BGP_ETRUNC; }
        Piece b{}; uint64_t next = 0;
        BGP_TRY(block_at(mf.p, mf.n, off, b, &next, c->last_error));
        b.skip = 0; b.keep = b.isize;
        if (inflate_piece(mf.p, b, blk.data(), scratch.data()) < 0) { c->last_error = "zlib: corrupted BGZF block"; return BGP_EINVAL; }
        if (skip < b.isize) {
            const uint32_t take = (uint32_t)std::min<uint64_t>(b.isize - skip, need - got.size() > 0 && got.size() >= 4 ? need - got.size() : b.isize - skip);
            segs.push_back(bgp_ctx::StageSeg{got.size(), take, off, skip});
            got.insert(got.end(), blk.data() + skip, blk.data() + skip + take);
        }
        if (got.size() >= 4 && need == 4) {
            int32_t bs; memcpy(&bs, got.data(), 4);
            if (bs < 32) { c->last_error = "bad block_size at the given virtual offset"; return BGP_EINVAL; }
            need = 4 + (uint64_t)(uint32_t)bs;
        }
        skip = 0; off = next;
    }
    got.resize(need);
    BGP_TRY(c->h_stage.reserve_keep(c->h_stage_len + need + 16, c->h_stage_len));
    memcpy(c->h_stage.as<uint8_t>() + c->h_stage_len, got.data(), need);
    for (auto s : segs) {
        if (s.staged >= need) break;
        s.len = (uint32_t)std::min<uint64_t>(s.len, need - s.staged);
        s.staged += c->h_stage_len;
        c->stage_map.push_back(s);
    }
    c->h_stage_len += need;
    return BGP_OK;
}
