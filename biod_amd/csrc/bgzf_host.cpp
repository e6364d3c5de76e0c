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
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>

#include <atomic>
#include <cstdio>
#include <thread>

namespace {

struct Blk { uint64_t coff; uint32_t cdata_off, cdata_len, crc, isize; uint64_t uoff; };

int parse_blocks(const uint8_t* f, uint64_t n, std::vector<Blk>& out, std::string& err) {
    uint64_t off = 0, uoff = 0;
    while (off < n) {
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

int inflate_block(const uint8_t* f, const Blk& b, uint8_t* dst) {         // block.d:127-216
    if (b.isize == 0) return BGP_OK;
    z_stream zs; memset(&zs, 0, sizeof zs);
    zs.next_in = const_cast<Bytef*>(f + b.coff + b.cdata_off); zs.avail_in = b.cdata_len;
    if (inflateInit2(&zs, -15) != Z_OK) return BGP_ENOMEM;
    zs.next_out = dst; zs.avail_out = b.isize;
    const int e = inflate(&zs, Z_FINISH);
    const uLong produced = zs.total_out;
    inflateEnd(&zs);
    if (e != Z_STREAM_END || produced != b.isize) return BGP_EINVAL;
    if ((uint32_t)crc32(0, dst, b.isize) != b.crc) return BGP_EINVAL;
    return BGP_OK;
}

}  // namespace

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
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { c->last_error = std::string("cannot open ") + path; return BGP_EINVAL; }
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); c->last_error = "fstat failed"; return BGP_EINVAL; }
    const size_t fsz = (size_t)sb.st_size;
    const uint8_t* file = fsz ? (const uint8_t*)mmap(nullptr, fsz, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    close(fd);
    if (fsz && file == (const uint8_t*)MAP_FAILED) { c->last_error = "mmap failed"; return BGP_EINVAL; }
    struct Unmap { const uint8_t* p; size_t n; ~Unmap() { if (p && n) munmap((void*)p, n); } } unmap{file, fsz};

    std::vector<Blk> blocks;
    int rc = parse_blocks(file, fsz, blocks, c->last_error);
    if (rc < 0) return rc;
    const uint64_t total = blocks.empty() ? 0 : blocks.back().uoff + blocks.back().isize;

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
    if (head.size() > q) memcpy(dst, head.data() + q, head.size() - (size_t)q);       // tail of the last header block
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads < 1) n_threads = 1;
    std::atomic<size_t> next{jh}; std::atomic<int> status{BGP_OK};
    auto worker = [&]() {
        for (;;) {
            const size_t j = next.fetch_add(1);
            if (j >= blocks.size() || status.load() < 0) return;
            const int e = inflate_block(file, blocks[j], dst + (blocks[j].uoff - q));
            if (e < 0) status.store(e);
        }
    };
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>((size_t)n_threads, blocks.size() - jh + 1);
    for (int t = 1; t < nt; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
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
