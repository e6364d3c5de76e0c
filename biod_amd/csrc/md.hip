// md.hip -- reference_base from MD tags on the device.
//
// Reference semantics: PileupRangeUsingMdTag (bio/std/hts/bam/pileup.d:522-654) keeps a "chunk" of
// reconstructed reference bases, dna(provider) (bio/std/hts/bam/md/reconstruct.d:38-214, MD parsed by
// mdOperations, bio/std/hts/bam/md/parse.d:13-142), and replaces it -- when it runs empty -- by the
// chunk of the best candidate seen since (add(), pileup.d:563-596).  The candidate chain is sequential
// per reference run, but only at EVENT granularity: between two chunk fetches nothing depends on read
// order except "first arg-max of end_position among the reads that pass add()'s filter", and the chunk
// cursor moves one base per visited column, i.e. in closed form.
//
//   k6_md_len   one lane per record : length of dna(read) and its "would hit an empty query" flag
//   k6_chain    one wave per reference run : the event loop (fetch to fetch), emits provider intervals
//   k6_fill     one lane per interval : dna(provider)[c0 ...] written to the dense reference_base array
//
// The event-level restatement was validated against the column-by-column oracle on random streams
// (gaps, N ops, missing / truncated / over-long MD, several references, both skip_zero_coverage modes)
// before it was written in HIP; tests/test_gpu_parity.py repeats that comparison through the ABI.
#include "common.h"

#include <algorithm>
#include <cstdio>

namespace {

// ---------------------------------------------------------------------------------------------- MD parsing
// mdOperations(md): a two-ended cache over the string (front parsed forwards, back parsed backwards)
// behind filterBidirectional!(not zero-match)  (md/parse.d:13-142).  Only forward iteration is used by
// dna(read), but the backward pre-parse decides where the forward parse stops, so it is kept literally.

struct MdOp { uint32_t type; /* 0 match, 1 mismatch, 2 deletion */ uint32_t match; uint32_t ch; uint32_t del_off, del_len; };
struct MdRes { const uint8_t* md; uint32_t b, e; MdOp front, back; int rem; };

__device__ __forceinline__ bool is_digit_(uint8_t c) { return c >= '0' && c <= '9'; }
__device__ __forceinline__ bool is_upper_(uint8_t c) { return c >= 'A' && c <= 'Z'; }

// Base16(c).asCharacter: char -> nt16 code -> char (bio/core/base.d:42-62,85; md/operation.d:92-97)
__device__ __forceinline__ uint8_t base16_norm(uint8_t c) {
    if (c >= '0' && c <= '3') return (uint8_t)"ACGT"[c - '0'];
    if (c == '=') return '=';
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 'a' + 'A');
    switch (c) {
        case 'A': case 'C': case 'M': case 'G': case 'R': case 'S': case 'V': case 'T':
        case 'W': case 'Y': case 'H': case 'K': case 'D': case 'B': case 'N': return c;
        default: return 'N';
    }
}

__device__ __forceinline__ uint32_t to_uint_(const uint8_t* p, uint32_t n) { uint64_t v = 0; for (uint32_t i = 0; i < n; ++i) v = v * 10 + (uint64_t)(p[i] - '0'); return (uint32_t)v; }

__device__ bool md_cache_front(MdRes& r) {                      // md/parse.d:85-109
    if (r.b >= r.e) return false;
    const uint8_t* m = r.md;
    MdOp o{0, 0, 0, 0, 0};
    if (m[r.b] == '^') {
        r.b++;
        uint32_t len = 0;
        while (r.b + len < r.e && is_upper_(m[r.b + len])) ++len;
        o.type = 2; o.del_off = r.b; o.del_len = len; r.b += len;
    } else if (is_digit_(m[r.b])) {
        uint32_t len = 0;
        while (r.b + len < r.e && is_digit_(m[r.b + len])) ++len;
        o.type = 0; o.match = to_uint_(m + r.b, len); r.b += len;
    } else { o.type = 1; o.ch = base16_norm(m[r.b]); r.b++; }
    r.front = o;
    return true;
}

__device__ bool md_cache_back(MdRes& r) {                       // md/parse.d:111-131
    if (r.b >= r.e) return false;
    const uint8_t* m = r.md;
    MdOp o{0, 0, 0, 0, 0};
    if (is_digit_(m[r.e - 1])) {
        uint32_t len = 0;
        while (r.e - len > r.b && is_digit_(m[r.e - 1 - len])) ++len;
        o.type = 0; o.match = to_uint_(m + r.e - len, len); r.e -= len;
    } else if (r.e - r.b == 1 || is_digit_(m[r.e - 2])) {
        o.type = 1; o.ch = base16_norm(m[r.e - 1]); r.e -= 1;
    } else {
        uint32_t len = 0;
        while (r.e - len > r.b && m[r.e - 1 - len] != '^') ++len;
        o.type = 2; o.del_off = r.e - len; o.del_len = len;
        r.e = (r.e - len > r.b) ? r.e - len - 1 : r.b;
    }
    r.back = o;
    return true;
}

__device__ __forceinline__ bool md_empty(const MdRes& r) { return r.rem == 0; }
__device__ void md_pop_front(MdRes& r) {                        // md/parse.d:57-69
    if (r.b >= r.e) { if (r.rem == 255) { r.front = r.back; r.rem = 1; } else r.rem = 0; }
    else if (!md_cache_front(r)) r.rem = 0;
}
__device__ void md_pop_back(MdRes& r) {                         // md/parse.d:71-83
    if (r.b >= r.e) { if (r.rem == 255) { r.back = r.front; r.rem = 1; } else r.rem = 0; }
    else if (!md_cache_back(r)) r.rem = 0;
}
__device__ __forceinline__ bool md_zero(const MdOp& o) { return o.type == 0 && o.match == 0; }
__device__ void mdf_init(MdRes& r, const uint8_t* md, uint32_t n) {   // md/parse.d:23-33 + filterBidirectional ctor
    r.md = md; r.b = 0; r.e = n; r.rem = 255;
    if (!md_cache_front(r)) r.rem = 0;
    else if (!md_cache_back(r)) { r.back = r.front; r.rem = 1; }
    while (!md_empty(r) && md_zero(r.front)) md_pop_front(r);
    while (!md_empty(r) && md_zero(r.back)) md_pop_back(r);
}
__device__ void mdf_pop_front(MdRes& r) { do { md_pop_front(r); } while (!md_empty(r) && md_zero(r.front)); }

// ---------------------------------------------------------------------------------------------- tag lookup
// BamRead.opIndex("MD") (read.d:1070-1087) with skipValue (read.d:1219-1230, tagvalue.d:78-119).
// Returns 1 found (a Z/H string: *val, *len), 0 absent or not a string, -1 malformed aux data.
__device__ int find_md(const uint8_t* s, const bgp_record& r, const uint8_t** val, uint32_t* len) {
    const uint32_t block = ld_u32(s + r.rec_off);
    const uint64_t end = r.rec_off + 4 + block;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    const uint64_t tags = r.rec_off + 36 + l_name + 4ull * n_cig + (uint64_t)((r.l_seq + 1) / 2) + (uint64_t)r.l_seq;
    if (tags > end) return -1;
    const uint8_t* t = s + tags;
    const uint64_t tl = end - tags;
    if (tl < 4) return 0;
    uint64_t off = 0;
    while (off + 1 < tl) {
        const bool match = t[off] == 'M' && t[off + 1] == 'D';
        off += 2;
        if (off >= tl) return -1;
        const uint8_t ty = t[off++];
        if (match) {
            if (ty != 'Z' && ty != 'H') return 0;
            uint32_t n = 0;
            while (off + n < tl && t[off + n] != 0) ++n;
            *val = t + off; *len = n;
            return 1;
        }
        if (ty == 'Z' || ty == 'H') { while (off < tl && t[off] != 0) ++off; if (off >= tl) return -1; ++off; }
        else if (ty == 'B') {
            if (off + 5 > tl) return -1;
            const uint8_t et = t[off];
            const int es = (et == 'c' || et == 'C') ? 1 : (et == 's' || et == 'S') ? 2 : (et == 'i' || et == 'I' || et == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += 5 + (uint64_t)es * ld_u32(t + off + 1);
        } else {
            const int es = (ty == 'A' || ty == 'c' || ty == 'C') ? 1 : (ty == 's' || ty == 'S') ? 2 : (ty == 'i' || ty == 'I' || ty == 'f') ? 4 : -1;
            if (es < 0) return -1;
            off += (uint64_t)es;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- dna(read)
// The query bases dna(read) draws from: SEQ under the M/=/X ops only (md/reconstruct.d:58-96,199).
struct QIter { const uint8_t* cg; const uint8_t* sq; uint32_t n_cig, op, in_op, q; bool empty; };
__device__ void q_settle(QIter& it) {
    while (it.op < it.n_cig) {
        const uint32_t raw = ld_u32(it.cg + 4ull * it.op);
        if (cig_qc(raw)) {
            if (cig_rc(raw) && it.in_op < (raw >> 4)) return;
            if (!cig_rc(raw)) it.q += raw >> 4;
        }
        it.op++; it.in_op = 0;
    }
    it.empty = true;
}
__device__ __forceinline__ uint8_t q_front(const QIter& it) {
    const uint32_t byte = it.sq[it.q >> 1];
    const uint32_t code = (it.q & 1u) ? (byte & 0xFu) : (byte >> 4);
    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;        // "=ACMGRSVTWYHKDBN", base.d:85
    return (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
}
__device__ __forceinline__ void q_pop(QIter& it) { it.q++; it.in_op++; q_settle(it); }

// Walks dna(read) (md/reconstruct.d:98-182).  Chars with index in [from, from+count) are stored to out
// (out may be null).  Returns the number of chars the lazy range yields before it ends; *bad is set when
// the range would stall on an empty query (the oracle's `bad`), *err on malformed aux data.
__device__ uint32_t dna_walk(const uint8_t* s, const bgp_record& r, uint32_t from, uint32_t count, uint8_t* out, bool* bad, bool* err) {
    *bad = false; *err = false;
    const uint8_t* md = nullptr; uint32_t mdn = 0;
    const int f = find_md(s, r, &md, &mdn);
    if (f < 0) { *err = true; return 0; }
    if (f == 0) mdn = 0;
    const uint32_t l_name = r.bin_mq_nl & 0xFFu, n_cig = r.flag_nc & 0xFFFFu;
    QIter q; q.cg = s + r.rec_off + 36 + l_name; q.sq = q.cg + 4ull * n_cig; q.n_cig = n_cig; q.op = 0; q.in_op = 0; q.q = 0; q.empty = false;
    q_settle(q);
    MdRes m;
    mdf_init(m, md, mdn);
    uint32_t n = 0;
    if (md_empty(m)) return 0;
    MdOp cur = m.front; mdf_pop_front(m);
    uint32_t del_i = 0;
    const uint32_t stop = out ? from + count : 0xFFFFFFFFu;
    auto put = [&](uint8_t c) { if (out && n >= from && n < stop) out[n - from] = c; ++n; };
    for (;;) {
        if (out && n >= stop) return n;
        if (cur.type == 1) {
            put((uint8_t)cur.ch);
            if (q.empty) { *bad = true; return n; }
            q_pop(q);
        } else if (cur.type == 0) {
            if (q.empty) { *bad = true; return n; }
            put(q_front(q));
            --cur.match;
            q_pop(q);
            if (cur.match != 0) continue;
        } else {
            if (cur.del_len == 0) { *bad = true; return n; }
            put(base16_norm(md[cur.del_off + del_i]));
            if (++del_i < cur.del_len) continue;
            del_i = 0;
        }
        if (md_empty(m)) return n;
        cur = m.front; mdf_pop_front(m);
    }
}

constexpr uint32_t MD_BAD = 1u << 31;

__global__ __launch_bounds__(256) void k6_md_len(const uint8_t* s, const bgp_record* rec, uint64_t n, uint32_t* md_len, uint32_t* status) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bgp_record r = rec[i];
    uint32_t v = 0;
    if (r.end_pos != r.pos) {
        bool bad, err;
        const uint32_t L = dna_walk(s, r, 0, 0, nullptr, &bad, &err);
        if (err) atomicOr(&status[0], ST_EINVAL);
        v = (L & ~MD_BAD) | (bad ? MD_BAD : 0u);
    }
    md_len[i] = v;
}

// debug/test hook: dna(read) of one record into a host-visible buffer
__global__ void k6_dna_one(const uint8_t* s, const bgp_record* rec, uint64_t i, uint8_t* out, uint32_t cap, uint32_t* len_bad) {
    bool bad, err, b2, e2;
    const uint32_t L = dna_walk(s, rec[i], 0, 0, nullptr, &bad, &err);        // full length (no early stop)
    if (cap) dna_walk(s, rec[i], 0, cap, out, &b2, &e2);
    len_bad[0] = L; len_bad[1] = (bad ? 1u : 0u) | (err ? 2u : 0u);
}

}  // namespace

int k6_md_lengths(bgp_ctx* c) {
    const uint64_t n = c->n_records;
    BGP_TRY(c->d_md_len.reserve((n + 1) * 4));
    if (n == 0) return BGP_OK;
    BGP_HIP(hipMemsetAsync(c->d_status.p, 0, 64, c->stream));
    TimedRegion t(c, "k6_md_len", n * 32);
    hipLaunchKernelGGL(k6_md_len, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), n, c->d_md_len.as<uint32_t>(), c->d_status.as<uint32_t>());
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}

int k6_dna_of_record(bgp_ctx* c, uint64_t i, uint8_t* host_out, uint32_t cap, uint32_t* len, uint32_t* flags) {
    if (i >= c->n_records) return BGP_EINVAL;
    BGP_TRY(c->d_entries.reserve((size_t)cap + 64));
    uint8_t* d = c->d_entries.as<uint8_t>();
    uint32_t* lb = (uint32_t*)(d + ((cap + 15) & ~15u));
    BGP_TRY(c->d_entries.reserve((size_t)((cap + 15) & ~15u) + 16));
    d = c->d_entries.as<uint8_t>(); lb = (uint32_t*)(d + ((cap + 15) & ~15u));
    hipLaunchKernelGGL(k6_dna_one, dim3(1), dim3(1), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), i, d, cap, lb);
    uint32_t h[2] = {0, 0};
    BGP_HIP(hipMemcpyAsync(h, lb, 8, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
    const uint32_t ncopy = std::min(h[0], cap);
    if (ncopy) BGP_HIP(hipMemcpy(host_out, d, ncopy, hipMemcpyDeviceToHost));
    *len = h[0]; *flags = h[1];
    return BGP_OK;
}

int k6_reference_bases(bgp_ctx* c, int skip_zero, uint32_t run_first, uint32_t run_end, uint64_t first_record) {
    (void)c; (void)skip_zero; (void)run_first; (void)run_end; (void)first_record;
    return BGP_EMD;     // stage 2
}
