// decode.hip -- kernel K2: 4-bit SEQ -> chars, QUAL copy, CIGAR copy into dense arenas.
//
// Reference semantics: SequenceResult.opIndex (bio/std/hts/bam/read.d:364-383: base i is the high
// nibble of byte i>>1 when i is even, else the low nibble) rendered through Base.asCharacter
// (bio/core/base.d:32,85 "=ACMGRSVTWYHKDBN"); base_qualities (read.d:468-470); cigar (read.d:225-228).
//
// MI355X design (HBM-bound byte shuffling, no MFMA).  A workgroup owns 256 consecutive records and
// therefore one contiguous byte range of each output arena.  Work is indexed by OUTPUT position:
// a lane produces one 16-byte-aligned 16-byte chunk of the seq arena (and the same chunk of the qual
// arena) per step, so every store is a full coalesced dwordx4 and a wave writes 1 KiB contiguous.
// The record owning a chunk is found by binary search over the workgroup's 257 arena offsets held in
// LDS; source bytes are read straight from the record stream with unaligned 8/16-byte loads
// (adjacent lanes read adjacent bytes, so the loads coalesce too).  Chunks that straddle a record
// boundary (about 1 in 10 for 150-base reads) take a byte-walking path.
#include "common.h"

namespace {

constexpr int RB = 256;   // records per workgroup

struct DecArgs {
    const uint8_t* s; uint64_t stream_len; const bgp_record* rec; const uint64_t* seq_off; const uint64_t* cigar_off; uint64_t n;
    uint8_t* seq; uint8_t* qual; uint32_t* cigar;
};

__device__ __forceinline__ uint8_t nt16_char(uint32_t code) {      // base.d:85
    // "=ACMGRSV" "TWYHKDBN" as little-endian u64 constants
    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;
    return (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
}

struct RecInfo { uint64_t cig_ptr, seq_ptr, qual_ptr; };

// largest j in [0, cnt) with off[j] <= b (caller guarantees off[0] <= b < off[cnt]); `g` is a guess
__device__ __forceinline__ int find_rec(const uint64_t* off /*LDS, cnt+1*/, int cnt, uint64_t b, int g) {
    g = min(max(g, 0), cnt - 1);
    if (off[g] <= b) {
        if (b < off[g + 1]) return g;
        if (g + 2 <= cnt && b < off[g + 2]) return g + 1;
    } else if (g > 0 && off[g - 1] <= b) return g - 1;
    int lo = 0, hi = cnt;
    while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (off[m] <= b) lo = m; else hi = m;
    }
    return lo;
}

// 16 consecutive bases, as chars, of the 4-bit stream whose nibble `odd` (0/1) of byte p[0] comes first.
// read.d:364-383 (even index = high nibble) + base.d:85 ("=ACMGRSVTWYHKDBN"), 4 bases per 32-bit op:
// swap the nibbles of every byte (stream order becomes little-endian nibble order), funnel-shift the
// odd start away, spread 4 nibbles into 4 bytes, look the bytes up with two v_perm_b32.
__device__ __forceinline__ uint4 expand16(const uint8_t* p, uint32_t odd) {
    uint2 w; __builtin_memcpy(&w, p, 8);                 // (non-temporal LOADS here cost 3 ms: neighbouring lanes' unaligned reads share lines)
    const uint32_t e = p[8];
    auto swp = [](uint32_t x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); };
    const uint32_t s0 = swp(w.x), s1 = swp(w.y), s2 = swp(e);
    const uint32_t t0 = __builtin_amdgcn_alignbit(s1, s0, odd * 4), t1 = __builtin_amdgcn_alignbit(s2, s1, odd * 4);
    auto chars4 = [](uint32_t h) {                       // 4 nibbles in h[15:0] -> 4 chars
        uint32_t z = (h | (h << 8)) & 0x00FF00FFu;
        z = (z | (z << 4)) & 0x0F0F0F0Fu;
        const uint32_t sel = z & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(0x56535247u, 0x4D43413Du, sel);   // "=ACM" "GRSV"
        const uint32_t hi = __builtin_amdgcn_perm(0x4E42444Bu, 0x48595754u, sel);   // "TWYH" "KDBN"
        const uint32_t m = ((z >> 3) & 0x01010101u) * 0xFFu;
        return (lo & ~m) | (hi & m);
    };
    return make_uint4(chars4(t0 & 0xFFFFu), chars4(t0 >> 16), chars4(t1 & 0xFFFFu), chars4(t1 >> 16));
}

// first n bytes from a, the rest from b
__device__ __forceinline__ uint4 merge16(uint4 a, uint4 b, uint32_t n) {
    auto m = [n](uint32_t d) { const int k = (int)n - 4 * (int)d; return k >= 4 ? 0xFFFFFFFFu : k <= 0 ? 0u : (1u << (8 * k)) - 1u; };
    const uint32_t m0 = m(0), m1 = m(1), m2 = m(2), m3 = m(3);
    return make_uint4((a.x & m0) | (b.x & ~m0), (a.y & m1) | (b.y & ~m1), (a.z & m2) | (b.z & ~m2), (a.w & m3) | (b.w & ~m3));
}

// Unpacks the `cnt` consecutive records whose arena offsets and section pointers sit in LDS.
template <bool SEQ, bool QUAL, bool CIG>
__device__ __forceinline__ void unpack_group(const DecArgs& a, int cnt, const uint64_t* s_seq_off, const uint64_t* s_cig_off, const RecInfo* s_info) {
    if (CIG) {
        const uint64_t c0 = s_cig_off[0], c1 = s_cig_off[cnt];
        const float inv = c1 > c0 ? (float)cnt / (float)(c1 - c0) : 0.f;
        for (uint64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
            const int j = find_rec(s_cig_off, cnt, i, (int)((float)(i - c0) * inv));
            a.cigar[i] = ld_u32(a.s + s_info[j].cig_ptr + 4 * (i - s_cig_off[j]));
        }
    }
    if (!SEQ && !QUAL) return;

    const uint64_t b0 = s_seq_off[0], b1 = s_seq_off[cnt];
    if (b1 == b0) return;
    const float inv = (float)cnt / (float)(b1 - b0);
    const uint64_t ch0 = b0 >> 4, ch1 = (b1 + 15) >> 4;
    for (uint64_t ch = ch0 + threadIdx.x; ch < ch1; ch += blockDim.x) {
        const uint64_t B = ch << 4;
        const uint64_t lo = max(B, b0), hi = min(B + 16, b1);     // bytes of this chunk this workgroup owns
        int j = find_rec(s_seq_off, cnt, lo, (int)((float)(lo - b0) * inv));
        const uint64_t q0 = lo - s_seq_off[j];
        const uint64_t end_j = s_seq_off[j + 1];
        bool done = false;
        if (lo == B && hi == B + 16) {
            const uint32_t n1 = (uint32_t)min((uint64_t)16, end_j - B);           // bytes of this chunk that belong to record j
            int j2 = j + 1;
            if (n1 < 16) while (j2 < cnt && s_seq_off[j2 + 1] == s_seq_off[j2]) ++j2;     // skip empty sequences
            const bool two = n1 < 16;
            const bool fits = !two || (j2 < cnt && s_seq_off[j2 + 1] - s_seq_off[j2] >= 16 - n1);
            const uint64_t p1 = s_info[j].seq_ptr + (q0 >> 1), g1 = s_info[j].qual_ptr + q0;
            const bool safe = p1 + 16 <= a.stream_len && g1 + 16 <= a.stream_len;
            if (fits && safe) {
                uint4 sv, qv;
                if (SEQ) sv = expand16(a.s + p1, (uint32_t)q0 & 1u);
                if (QUAL) __builtin_memcpy(&qv, a.s + g1, 16);
                if (two) {                                                         // chunk straddles records j | j2
                    const int64_t nq = -(int64_t)n1;                               // stream position of out[0] relative to record j2
                    if (SEQ) sv = merge16(sv, expand16(a.s + (s_info[j2].seq_ptr + (nq >> 1)), (uint32_t)nq & 1u), n1);
                    if (QUAL) { uint4 q2; __builtin_memcpy(&q2, a.s + (s_info[j2].qual_ptr - n1), 16); qv = merge16(qv, q2, n1); }
                }
                if (SEQ) nt_store16(a.seq + B, sv);
                if (QUAL) nt_store16(a.qual + B, qv);
                done = true;
            }
        }
        if (!done) {
            // workgroup-boundary chunk, three or more records inside 16 bytes, or the very end of the stream
            for (uint64_t b = lo; b < hi; ++b) {
                while (b >= s_seq_off[j + 1]) ++j;                 // skips empty sequences too
                const uint64_t q = b - s_seq_off[j];
                if (SEQ) {
                    const uint32_t byte = a.s[s_info[j].seq_ptr + (q >> 1)];
                    a.seq[b] = nt16_char((q & 1) ? (byte & 0xFu) : (byte >> 4));
                }
                if (QUAL) a.qual[b] = a.s[s_info[j].qual_ptr + q];
            }
        }
    }
}

template <bool SEQ, bool QUAL, bool CIG>
__global__ __launch_bounds__(256) void k2_unpack(DecArgs a) {
    __shared__ uint64_t s_seq_off[RB + 1];
    __shared__ uint64_t s_cig_off[RB + 1];
    __shared__ RecInfo s_info[RB];
    const uint64_t r0 = (uint64_t)blockIdx.x * RB;
    const int cnt = (int)min((uint64_t)RB, a.n - r0);
    for (int j = threadIdx.x; j <= cnt; j += blockDim.x) {
        s_seq_off[j] = a.seq_off[r0 + j];
        s_cig_off[j] = a.cigar_off[r0 + j];
        if (j < cnt) {
            const bgp_record r = a.rec[r0 + j];
            RecInfo ri;
            ri.cig_ptr = r.rec_off + 36 + (r.bin_mq_nl & 0xFFu);                    // read.d:988-990
            ri.seq_ptr = ri.cig_ptr + 4ull * (r.flag_nc & 0xFFFFu);                 // read.d:992-994
            ri.qual_ptr = ri.seq_ptr + (uint64_t)((r.l_seq + 1) / 2);               // read.d:996-998
            s_info[j] = ri;
        }
    }
    __syncthreads();
    unpack_group<SEQ, QUAL, CIG>(a, cnt, s_seq_off, s_cig_off, s_info);
}

// ---------------------------------------------------------------------------------------------- K1 + K2 in one pass
// When a decode follows the framing directly, the record table does not need its own pass over the stream: k0_chase
// left every record's offset (relative to its segment) in a side array, the scans gave every segment its base record
// index and arena offsets, so a workgroup that owns G_SEG consecutive segments can compute its records' table rows
// with one lane per record (header + CIGAR are in the lines the unpack is about to read anyway), turn l_seq / n_cigar
// into arena offsets with a block scan, write the rows coalesced, and unpack -- the 29 GB stream is read once
// instead of twice.
constexpr int G_SEG = 8;

struct FusedArgs {
    DecArgs d;
    const uint64_t* entry; const uint64_t* rec_base; const uint64_t* seq_base; const uint64_t* cig_base; const uint32_t* side;
    uint32_t side_cap, seg_bytes; uint64_t n_seg, sum_lseq, sum_ncig;
    bgp_record* rec_out; uint64_t* seq_off_out; uint64_t* cig_off_out; uint4* cig4_out;
    unsigned long long* tilemax;      // K3's keys, maximum per 2048-record tile (two u64 per tile), or null
};

// exclusive prefix sums of two per-thread values over the 256 threads of the block + the block totals
__device__ __forceinline__ void block_excl2(uint64_t v0, uint64_t v1, uint64_t* e0, uint64_t* e1, uint64_t* t0, uint64_t* t1, uint64_t (*ws)[4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t i0 = v0, i1 = v1;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o0 = __shfl_up(i0, d, 64), o1 = __shfl_up(i1, d, 64);
        if (lane >= d) { i0 += o0; i1 += o1; }
    }
    if (lane == 63) { ws[0][w] = i0; ws[1][w] = i1; }
    __syncthreads();
    uint64_t c0 = 0, c1 = 0, s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (k < w) { c0 += ws[0][k]; c1 += ws[1][k]; } s0 += ws[0][k]; s1 += ws[1][k]; }
    *e0 = c0 + i0 - v0; *e1 = c1 + i1 - v1; *t0 = s0; *t1 = s1;
    __syncthreads();
}

template <bool SEQ, bool QUAL, bool CIG>
__global__ __launch_bounds__(256) void k12_fused(FusedArgs a) {
    __shared__ uint64_t s_seq_off[RB + 1];
    __shared__ uint64_t s_cig_off[RB + 1];
    __shared__ RecInfo s_info[RB];
    __shared__ uint64_t s_rb[G_SEG + 1];
    __shared__ uint64_t s_ws[2][4];
    const uint64_t k0 = (uint64_t)blockIdx.x * G_SEG;
    const int g_n = (int)min((uint64_t)G_SEG, a.n_seg - k0);
    if ((int)threadIdx.x <= g_n) s_rb[threadIdx.x] = a.rec_base[k0 + threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.seq_off_out[a.d.n] = a.sum_lseq; a.cig_off_out[a.d.n] = a.sum_ncig; }
    __syncthreads();
    const uint64_t r_lo = s_rb[0], r_hi = s_rb[g_n];
    uint64_t carry_seq = a.seq_base[k0], carry_cig = a.cig_base[k0];
    for (uint64_t b0 = r_lo; b0 < r_hi; b0 += RB) {
        const int cnt = (int)min((uint64_t)RB, r_hi - b0);
        const uint64_t i = b0 + threadIdx.x;
        const bool valid = (int)threadIdx.x < cnt;
        bgp_record rec{};
        uint64_t lseq = 0, ncig = 0;
        uint4 c4v = make_uint4(0, 0, 0, 0);                          // the record's first four CIGAR ops (the planner of K4 reads them from here)
        if (valid) {
            int g = 0;
#pragma unroll
            for (int k = 1; k < G_SEG; ++k) if (k < g_n && s_rb[k] <= i) g = k;
            const uint64_t seg = k0 + g;
            const uint64_t off = seg * a.seg_bytes + a.side[seg * a.side_cap + (i - s_rb[g])];
            const uint8_t* p = a.d.s + off;
            // few wide loads instead of many dword loads: the lanes' records are ~300 bytes apart, so every load
            // instruction costs the address coalescer 64 separate lines whatever its width
            uint4 h; __builtin_memcpy(&h, p + 4, 16);                // ref_id, pos, bin_mq_nl, flag_nc: read.d:907-919
            rec.ref_id = (int32_t)h.x; rec.pos = (int32_t)h.y; rec.bin_mq_nl = h.z; rec.flag_nc = h.w;
            rec.l_seq = (int32_t)ld_u32(p + 20);                     // read.d:923
            rec.rec_off = off;
            const uint32_t l_name = rec.bin_mq_nl & 0xFFu, n_cig = rec.flag_nc & 0xFFFFu;
            uint32_t covered = 0;
            if (!((rec.flag_nc >> 16) & 0x4u) && n_cig <= BGP_LONG_CIGAR) {   // read.d:257-259; long CIGARs: k1_long_cigar afterwards
                const uint8_t* cg = p + 36 + l_name;                  // read.d:988-990
                uint32_t j0 = 0;
                if (n_cig && off + 36 + l_name + 16 <= a.d.stream_len) {       // the first four ops in one load
                    uint4 c4; __builtin_memcpy(&c4, cg, 16);
                    covered += cig_rc(c4.x) ? (c4.x >> 4) : 0u;
                    if (n_cig > 1) covered += cig_rc(c4.y) ? (c4.y >> 4) : 0u;
                    if (n_cig > 2) covered += cig_rc(c4.z) ? (c4.z >> 4) : 0u;
                    if (n_cig > 3) covered += cig_rc(c4.w) ? (c4.w >> 4) : 0u;
                    j0 = 4;
                    c4v = make_uint4(c4.x, n_cig > 1 ? c4.y : 0u, n_cig > 2 ? c4.z : 0u, n_cig > 3 ? c4.w : 0u);
                }
                for (uint32_t j = j0; j < n_cig; ++j) {                                                                                      // read.d:261
                    const uint32_t op = ld_u32(cg + 4 * j); covered += cig_rc(op) ? (op >> 4) : 0u;
                    if (j == 0) c4v.x = op; else if (j == 1) c4v.y = op; else if (j == 2) c4v.z = op; else if (j == 3) c4v.w = op;
                }
            }
            rec.end_pos = (int32_t)((uint32_t)rec.pos + covered);     // read.d:1382
            lseq = (uint64_t)(uint32_t)rec.l_seq; ncig = n_cig;
        }
        uint64_t es, ec, ts, tc;
        block_excl2(lseq, ncig, &es, &ec, &ts, &tc, s_ws);
        const uint64_t so = carry_seq + es, co = carry_cig + ec;
        if (valid) {
            a.rec_out[i] = rec; a.seq_off_out[i] = so; a.cig_off_out[i] = co;       // (plain stores: K2 reads these rows right after, non-temporal ones cost it 0.3 ms)
            a.cig4_out[i] = c4v;
        }
        if (a.tilemax) {            // what K3's first pass would compute from these rows: a wave covers 64 consecutive records, i.e. one or two tiles
            K3Key k{0, 0};
            if (valid) k = k3_key_of(rec.ref_id, rec.pos, rec.end_pos);
            const uint64_t tile = i / K3_TILE_RECORDS;
            const uint64_t t0 = __shfl(tile, 0, 64);
            K3Key m0 = tile == t0 ? k : K3Key{0, 0}, m1 = tile == t0 ? K3Key{0, 0} : k;
            for (int d = 32; d >= 1; d >>= 1) {
                const uint64_t e0 = __shfl_xor(m0.e, d, 64), p0 = __shfl_xor(m0.p, d, 64), e1 = __shfl_xor(m1.e, d, 64), p1 = __shfl_xor(m1.p, d, 64);
                m0.e = max(m0.e, e0); m0.p = max(m0.p, p0); m1.e = max(m1.e, e1); m1.p = max(m1.p, p1);
            }
            if ((threadIdx.x & 63) == 0) {
                if (m0.e) atomicMax(&a.tilemax[2 * t0], (unsigned long long)m0.e);
                if (m0.p) atomicMax(&a.tilemax[2 * t0 + 1], (unsigned long long)m0.p);
                if (m1.e) atomicMax(&a.tilemax[2 * (t0 + 1)], (unsigned long long)m1.e);
                if (m1.p) atomicMax(&a.tilemax[2 * (t0 + 1) + 1], (unsigned long long)m1.p);
            }
        }
        if (valid) {
            s_seq_off[threadIdx.x] = so; s_cig_off[threadIdx.x] = co;
            RecInfo ri;
            ri.cig_ptr = rec.rec_off + 36 + (rec.bin_mq_nl & 0xFFu);
            ri.seq_ptr = ri.cig_ptr + 4ull * (rec.flag_nc & 0xFFFFu);
            ri.qual_ptr = ri.seq_ptr + (uint64_t)((rec.l_seq + 1) / 2);
            s_info[threadIdx.x] = ri;
        }
        if ((int)threadIdx.x == cnt - 1 || (cnt == 0 && threadIdx.x == 0)) { s_seq_off[cnt] = carry_seq + ts; s_cig_off[cnt] = carry_cig + tc; }
        __syncthreads();
        unpack_group<SEQ, QUAL, CIG>(a.d, cnt, s_seq_off, s_cig_off, s_info);
        __syncthreads();
        carry_seq += ts; carry_cig += tc;
    }
}

}  // namespace

int k2_decode(bgp_ctx* c, uint32_t what) {
    const uint64_t n = c->n_records;
    if (what & BGP_FETCH_SEQ) BGP_TRY(c->d_seq.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_QUAL) BGP_TRY(c->d_qual.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_CIGAR) BGP_TRY(c->d_cigar.reserve(c->sum_ncig * 4 + 16));
    c->decoded_what |= what;
    if (n == 0) return BGP_OK;
    DecArgs a;
    a.s = c->d_stream; a.stream_len = c->stream_len; a.rec = c->d_rec.as<bgp_record>(); a.seq_off = c->d_seq_off.as<uint64_t>(); a.cigar_off = c->d_cigar_off.as<uint64_t>();
    a.n = n; a.seq = c->d_seq.as<uint8_t>(); a.qual = c->d_qual.as<uint8_t>(); a.cigar = c->d_cigar.as<uint32_t>();
    const unsigned blocks = (unsigned)((n + RB - 1) / RB);
    const bool S = what & BGP_FETCH_SEQ, Q = what & BGP_FETCH_QUAL, C = what & BGP_FETCH_CIGAR;
    uint64_t alg = n * 48;    // record table + two offsets read
    if (S) alg += (c->sum_lseq + 1) / 2 + c->sum_lseq;
    if (Q) alg += 2 * c->sum_lseq;
    if (C) alg += 8 * c->sum_ncig;
    TimedRegion t(c, "k2_unpack", alg);
#define LAUNCH(s_, q_, c_) hipLaunchKernelGGL((k2_unpack<s_, q_, c_>), dim3(blocks), dim3(256), 0, c->stream, a)
    if (S && Q && C) LAUNCH(true, true, true);
    else if (S && Q) LAUNCH(true, true, false);
    else if (S && C) LAUNCH(true, false, true);
    else if (Q && C) LAUNCH(false, true, true);
    else if (S) LAUNCH(true, false, false);
    else if (Q) LAUNCH(false, true, false);
    else if (C) LAUNCH(false, false, true);
#undef LAUNCH
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}

// K1 and K2 fused: used by bgp_decode_* when the record table has not been written yet (api.cpp).
int k12_fused_decode(bgp_ctx* c, uint32_t what) {
    const uint64_t n = c->n_records;
    c->table_pending = false;
    BGP_TRY(c->d_rec.reserve((n + 1) * sizeof(bgp_record)));
    BGP_TRY(c->d_seq_off.reserve((n + 1) * 8)); BGP_TRY(c->d_cigar_off.reserve((n + 1) * 8));
    BGP_TRY(c->d_cig4.reserve((n + 1) * 16)); c->cig4_ok = true;
    if (what & BGP_FETCH_SEQ) BGP_TRY(c->d_seq.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_QUAL) BGP_TRY(c->d_qual.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_CIGAR) BGP_TRY(c->d_cigar.reserve(c->sum_ncig * 4 + 16));
    c->decoded_what |= what;
    if (c->stream_len < 4) return BGP_OK;
    FusedArgs a;
    a.d.s = c->d_stream; a.d.stream_len = c->stream_len; a.d.rec = nullptr; a.d.seq_off = nullptr; a.d.cigar_off = nullptr; a.d.n = n;
    a.d.seq = c->d_seq.as<uint8_t>(); a.d.qual = c->d_qual.as<uint8_t>(); a.d.cigar = c->d_cigar.as<uint32_t>();
    a.entry = c->d_seg_entry.as<uint64_t>(); a.rec_base = c->d_seg_rec_base.as<uint64_t>(); a.seq_base = c->d_seg_seq_base.as<uint64_t>(); a.cig_base = c->d_seg_cig_base.as<uint64_t>();
    a.side = c->d_seg_side.as<uint32_t>(); a.side_cap = c->side_cap; a.seg_bytes = c->seg_bytes; a.n_seg = c->n_seg; a.sum_lseq = c->sum_lseq; a.sum_ncig = c->sum_ncig;
    a.rec_out = c->d_rec.as<bgp_record>(); a.seq_off_out = c->d_seq_off.as<uint64_t>(); a.cig_off_out = c->d_cigar_off.as<uint64_t>();
    a.cig4_out = c->d_cig4.as<uint4>();
    // K3's tile maxima on the way -- unless some record's end_position is finished later (CIGARs longer than the inline loop)
    a.tilemax = nullptr; c->tilemax_ok = false;
    if (!c->long_cigar) {
        const uint64_t nt = (n + K3_TILE_RECORDS - 1) / K3_TILE_RECORDS;
        BGP_TRY(c->d_tilemax.reserve((nt + 2) * 16));
        BGP_HIP(hipMemsetAsync(c->d_tilemax.p, 0, (nt + 2) * 16, c->stream));
        a.tilemax = (unsigned long long*)c->d_tilemax.p; c->tilemax_ok = true;
    }
    const unsigned blocks = (unsigned)((c->n_seg + G_SEG - 1) / G_SEG);
    const bool S = what & BGP_FETCH_SEQ, Q = what & BGP_FETCH_QUAL, C = what & BGP_FETCH_CIGAR;
    uint64_t alg = n * (36 + 32 + 16) + c->sum_ncig * 4;    // what k1_emit accounts for ...
    if (S) alg += (c->sum_lseq + 1) / 2 + c->sum_lseq;      // ... plus what k2_unpack accounts for, minus the table read back
    if (Q) alg += 2 * c->sum_lseq;
    if (C) alg += 8 * c->sum_ncig;
    {
        TimedRegion t(c, what ? "k12_fused" : "k1_rows", alg);
#define LAUNCH(s_, q_, c_) hipLaunchKernelGGL((k12_fused<s_, q_, c_>), dim3(blocks), dim3(256), 0, c->stream, a)
        if (S && Q && C) LAUNCH(true, true, true);
        else if (S && Q) LAUNCH(true, true, false);
        else if (S && C) LAUNCH(true, false, true);
        else if (Q && C) LAUNCH(false, true, true);
        else if (S) LAUNCH(true, false, false);
        else if (Q) LAUNCH(false, true, false);
        else if (C) LAUNCH(false, false, true);
        else LAUNCH(false, false, false);
#undef LAUNCH
        BGP_TRY(k1_finish_long_cigars(c));
    }
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}
