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
#include <cstdio>
#include <algorithm>

namespace {

constexpr int RB = 256;   // records per workgroup

struct DecArgs {
    const uint8_t* s; uint64_t stream_len; const bgp_record* rec; const uint64_t* seq_off; const uint64_t* cigar_off; uint64_t n;
    uint8_t* seq; uint8_t* qual; uint32_t* cigar;
    uint32_t rb;          // records per workgroup of k2_unpack (<= RB): fewer when records are long, so that the grid still fills the device
    bgp_record* rec_fix;  // k2_unpack: records with more than BGP_LONG_CIGAR ops get their end_position here, from the ops it copies (null: k1_long_cigar has done it)
};

__device__ __forceinline__ uint8_t nt16_char(uint32_t code) {      // base.d:85
    // "=ACMGRSV" "TWYHKDBN" as little-endian u64 constants
    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;
    return (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
}

struct RecInfo { uint64_t cig_ptr, seq_ptr, qual_ptr; uint32_t sum_cov, pad; };      // sum_cov: this record's basesCovered is summed while its CIGAR is copied (read.d:255-262)

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

// first n bytes from a, the rest from b (0 <= n <= 16): two 64-bit masks instead of four 32-bit ones with their clamps
__device__ __forceinline__ uint4 merge16(uint4 a, uint4 b, uint32_t n) {
    const unsigned long long lo = n >= 8 ? ~0ull : (1ull << (8u * n)) - 1ull;
    const unsigned long long hi = n <= 8 ? 0ull : n >= 16 ? ~0ull : (1ull << (8u * (n - 8u))) - 1ull;
    const uint32_t m0 = (uint32_t)lo, m1 = (uint32_t)(lo >> 32), m2 = (uint32_t)hi, m3 = (uint32_t)(hi >> 32);
    return make_uint4((a.x & m0) | (b.x & ~m0), (a.y & m1) | (b.y & ~m1), (a.z & m2) | (b.z & ~m2), (a.w & m3) | (b.w & ~m3));
}

// Unpacks the `cnt` consecutive records whose arena offsets and section pointers sit in LDS.
template <bool SEQ, bool QUAL, bool CIG>
__device__ __forceinline__ void unpack_group(const DecArgs& a, int cnt, const uint64_t* s_seq_off, const uint64_t* s_cig_off, const RecInfo* s_info, uint32_t* s_cov = nullptr) {
    if (CIG) {
        const uint64_t c0 = s_cig_off[0], c1 = s_cig_off[cnt];
        const float inv = c1 > c0 ? (float)cnt / (float)(c1 - c0) : 0.f;
        for (uint64_t ib = c0; ib < c1; ib += blockDim.x) {      // (a bound every thread agrees on: the wave reduction below is over all 64 lanes, none of them masked off)
            const uint64_t i = ib + threadIdx.x;
            const bool in = i < c1;
            int j = 0; uint32_t op = 0;
            if (in) {
                j = find_rec(s_cig_off, cnt, i, (int)((float)(i - c0) * inv));
                op = ld_u32(a.s + s_info[j].cig_ptr + 4 * (i - s_cig_off[j]));
                a.cigar[i] = op;
            }
            if (s_cov) {                                   // basesCovered of the long CIGARs on the way (a wave's ops are nearly always one record's: one LDS atomic a wave)
                const uint32_t cov = in && s_info[j].sum_cov && cig_rc(op) ? op >> 4 : 0u;
                const unsigned long long act = __ballot(in);
                if (act) {
                    const int l0 = __ffsll((long long)act) - 1;
                    const int j0 = __shfl(j, l0, 64);
                    if (__ballot(in && j != j0) == 0) {
                        uint32_t t = cov;                  // (0 in the lanes behind the last op)
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) t += (uint32_t)__shfl_xor((int)t, d, 64);
                        if (t && (int)(threadIdx.x & 63) == l0) atomicAdd(&s_cov[j0], t);
                    } else if (cov) atomicAdd(&s_cov[j], cov);
                }
            }
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
    __shared__ uint32_t s_cov[RB];
    const uint64_t r0 = (uint64_t)blockIdx.x * a.rb;
    const int cnt = (int)min((uint64_t)a.rb, a.n - r0);
    const bool fix = CIG && a.rec_fix != nullptr;
    for (int j = threadIdx.x; j <= cnt; j += blockDim.x) {
        s_seq_off[j] = a.seq_off[r0 + j];
        s_cig_off[j] = a.cigar_off[r0 + j];
        if (j < cnt) {
            const bgp_record r = a.rec[r0 + j];
            RecInfo ri;
            ri.cig_ptr = r.rec_off + 36 + (r.bin_mq_nl & 0xFFu);                    // read.d:988-990
            ri.seq_ptr = ri.cig_ptr + 4ull * (r.flag_nc & 0xFFFFu);                 // read.d:992-994
            ri.qual_ptr = ri.seq_ptr + (uint64_t)((r.l_seq + 1) / 2);               // read.d:996-998
            ri.sum_cov = fix && (r.flag_nc & 0xFFFFu) > BGP_LONG_CIGAR && !((r.flag_nc >> 16) & 0x4u) ? 1u : 0u; ri.pad = 0;      // read.d:257-259
            s_info[j] = ri;
            s_cov[j] = 0;
        }
    }
    __syncthreads();
    unpack_group<SEQ, QUAL, CIG>(a, cnt, s_seq_off, s_cig_off, s_info, fix ? s_cov : nullptr);
    if (fix) {
        __syncthreads();
        for (int j = threadIdx.x; j < cnt; j += blockDim.x) if (s_cov[j]) a.rec_fix[r0 + j].end_pos = a.rec_fix[r0 + j].pos + (int32_t)s_cov[j];      // read.d:1382: position + basesCovered (written, not added: doing it twice changes nothing)
    }
}

// ---------------------------------------------------------------------------------------------- K1 + K2 in one pass
// When a decode follows the framing directly, the record table does not need its own pass over the stream: k0_chase
// left every record's offset (relative to its segment) in a side array, the scans gave every segment its base record
// index and arena offsets, so a workgroup that owns G_SEG consecutive segments can compute its records' table rows
// with one lane per record (header + CIGAR are in the lines the unpack is about to read anyway), turn l_seq / n_cigar
// into arena offsets with a block scan, write the rows coalesced, and unpack -- the 29 GB stream is read once
// instead of twice.
constexpr int G_SEG = 8;
#ifndef BGP_K12_MINW
#define BGP_K12_MINW 4
#endif

struct FusedArgs {
    DecArgs d;
    const uint64_t* entry; const uint64_t* rec_base; const uint64_t* seq_base; const uint64_t* cig_base; const uint32_t* side;
    uint32_t side_cap, seg_bytes; int side16; uint64_t n_seg, sum_lseq, sum_ncig;
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
__global__ __launch_bounds__(256, BGP_K12_MINW) void k12_fused(FusedArgs a) {
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
            const uint64_t off = seg * a.seg_bytes + (a.side16 ? (uint32_t)reinterpret_cast<const uint16_t*>(a.side + seg * a.side_cap)[i - s_rb[g]] : a.side[seg * a.side_cap + (i - s_rb[g])]);
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


// ---------------------------------------------------------------------------------------------- K0 + K1 + K2 in ONE pass
// The decode by SURVEY 8(d)'s own byte count -- every byte of the stream in once, the table row and the arenas out
// once -- is what k_stream_decode does; K0 / K1 / K2 as separate kernels read the header lines of the stream three
// times (chase, rows, unpack) and its body once.
//
// A workgroup draws a ticket k and owns the records that START in stream bytes [k C, (k+1) C).  It copies
// [k C, k C + C + OV) into LDS with coalesced 16-byte loads and does everything else from there:
//   guess   the first offset in its chunk where a plausible record header is followed by two more (256 candidates per
//           step) -- exactly k0_guess's role: a guess costs time when wrong, never correctness;
//   chase   one lane hops `off += 4 + block_size` through LDS (~100 cycles a hop instead of an HBM round trip) and
//           leaves the records' offsets; all lanes then validate the records' sections (readrange.d:118-173 and the
//           bounds read.d's accessors rely on) and sum l_seq / n_cigar_op;
//   link    what the chunk found is published CONDITIONALLY -- "entered at g, I leave at x with n records, S bases,
//           K ops" -- as self-tagged 8-byte granules (relaxed agent-scope atomics: the data is the flag).  One wave
//           then looks back over its predecessors' descriptors, 64 at a time, down to the nearest chunk whose result is
//           final: a conditional result counts if the exit of the chunk before it IS its guess.  That yields the true
//           entry of this chunk and the number of records / bases / ops in front of it in one go (a decoupled
//           look-back scan whose operator carries the chain of record boundaries).  A chunk whose guess turns out
//           wrong chases again from the true entry (in LDS: cheap) and everybody behind it waits for its final
//           descriptor.  Chunk 0 enters at 0; by induction every final descriptor is the serial chain's.
//   emit    rows with one lane per record (fields and CIGAR from LDS, block scan for the arena offsets, coalesced
//           stores), then the unpack with one lane per 16-byte arena chunk as in k2_unpack, sources in LDS.
// A record that runs past the window (only a chunk's last one can) is unpacked from global memory by the whole
// workgroup.  Arena sizes are not known before the pass: the host sizes them from the first records and the pass
// reports an overflow instead of writing past a buffer (then it is run again with the exact sizes it found).
#ifndef BGP_K12_NT_ROWS
#define BGP_K12_NT_ROWS 1            // the per-record arrays leave with non-temporal stores like the arenas (nothing reads them before K3: k12_lds 13.23 -> 13.11 ms)
#endif
#ifndef BGP_SD_NW
#define BGP_SD_NW 8
#define BGP_SD_NS 4
#define BGP_SD_SC 8192
#define BGP_SD_MINW 8                          // four 512-thread workgroups per CU (40 KB of LDS each, <= 64 VGPRs): k12_lds 14.7 -> 13.5 ms against three
#endif
#ifndef BGP_SD_OV
#define BGP_SD_OV 2048
#endif
constexpr int SD_NW = BGP_SD_NW;                   // waves of a workgroup
constexpr int SD_NS = BGP_SD_NS;                   // sub-chunks framed side by side (one wave each; fewer sub-chunks: fewer guesses -- the kernel is bound by VALU issue -- but longer serial chases)
constexpr uint32_t SD_SC = BGP_SD_SC;              // bytes of a sub-chunk
constexpr uint32_t SD_C = SD_NS * SD_SC;           // stream bytes a workgroup owns at a time
constexpr uint32_t SD_OV = BGP_SD_OV;                   // how far past its chunk the LDS window reaches
constexpr uint32_t SD_WIN = SD_C + SD_OV;
constexpr int SD_T = 64 * SD_NW;
constexpr int SD_RB = 256;                         // records per emission round (rows: one lane each)
constexpr uint32_t SD_RMAX = SD_C / 36 + 2;        // a record is at least 36 bytes
constexpr uint32_t SD_SUBCAP = SD_SC / 36 + 2;
typedef unsigned long long u64;
// LDS map (bytes)
constexpr uint32_t L_ROFF = SD_WIN + 32;                              // u16[SD_RMAX]: where the chunk's records start (window offsets)
constexpr uint32_t L_TAB = (L_ROFF + 2 * SD_RMAX + 15) & ~15u;        // emission tables: soff u32[257] | coff u32[257] | info u16[3 x 256] -- before that, the sub-chunks' own offset lists
constexpr uint32_t L_SOFF = L_TAB, L_COFF = L_TAB + 1040, L_INFO = L_TAB + 2080;
[[maybe_unused]] constexpr uint32_t L_SUB = L_TAB;                                     // u16[SD_NS][SD_SUBCAP]
constexpr uint32_t L_MISC = L_INFO + 1536;                            // u64[84]
constexpr uint32_t L_TOTAL = L_MISC + 8 * 84;
static_assert(SD_WIN + 32 <= 65535 && SD_NS * SD_SUBCAP * 2 <= 2080 + 1536, "the sub-chunk lists overlay the emission tables");
static_assert(SD_C % 16 == 0 && SD_T <= 1024 && SD_RB <= SD_T, "geometry");
static_assert(2 * SD_RB <= SD_T, "plan rows are made by the SD_RB threads behind the table's (sd_emit): fewer waves would leave rows unwritten");
static_assert(BGP_SD_MINW * 4 / SD_NW * L_TOTAL <= 160 * 1024, "the workgroups the launch bounds promise must fit the LDS");
enum { M_TICKET = 0, M_FOUND, M_CNT, M_EXIT, M_ERR, M_BAD, M_LS, M_NC, M_FLAGS, M_ENT, M_PN, M_PLS, M_PNC, M_ABORT, M_TM = 16 /* 4 tile maxima */, M_WS = 20 /* 2 x 8 u64 of scan scratch */,
       M_SUBG = 36 /* SD_NS guesses */, M_SUBX = 44 /* exits */, M_SUBC = 52 /* records | status << 32 */, M_EFF = 60 /* records the chunk's chain takes from the sub-chunk */, M_SUBLS = 68 /* bases */, M_SUBNC = 76 /* CIGAR ops | a CIGAR longer than BGP_LONG_CIGAR << 63 */, M_N = 84 };
static_assert(SD_NS <= 8 && SD_NS <= SD_NW, "misc slots; a wave per sub-chunk");

struct SdArgs {
    const uint8_t* s; uint64_t len, n_chunks; int32_t ref_bound; int32_t tolerant;
    const uint64_t* hint;             // per chunk: smallest offset where the caller says a record starts (~0: no word), or null
    u64* desc;                        // per chunk 8 granules: P exit, n, bases, ops (inclusive) | A exit, guess|n|ops, bases | -
    u64* ctl;                         // [0] ticket [1] abort [2] overflow [3] records [4] bases [5] ops [6] where the chain stopped
    uint32_t* status;
    bgp_record* rec; uint64_t* seq_off; uint64_t* cig_off; uint4* cig4; unsigned long long* tilemax;
    uint8_t* seq; uint8_t* qual; uint32_t* cigar;
    uint64_t cap_n, cap_ls, cap_nc;
    uint4* prow; uint8_t* rcls;       // K4's plan rows and classes (common.h: bgp_plan_row), or null
};


// unaligned little-endian reads at byte offset o of the LDS window: the aligned dwords around it, funnel-shifted (v_alignbyte).  gfx950's DS
// unit does take unaligned addresses -- one ds_read each, fewer VALU instructions in a kernel whose SIMDs issue 80 % of their cycles -- but
// measured slower (14.2-14.3 against 13.4 ms: the unaligned accesses cost the LDS more than they save the VALU); BGP_WL_NATIVE=1 builds that.
#ifndef BGP_WL_NATIVE
#define BGP_WL_NATIVE 0
#endif
#if BGP_WL_NATIVE
__device__ __forceinline__ uint32_t wl32(const uint32_t* w, uint32_t o) { uint32_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(w) + o, 4); return v; }
__device__ __forceinline__ uint2 wl64(const uint32_t* w, uint32_t o) { uint2 v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(w) + o, 8); return v; }
__device__ __forceinline__ uint4 wl128(const uint32_t* w, uint32_t o) { uint4 v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(w) + o, 16); return v; }
struct R3 { uint32_t x, y, z; };
__device__ __forceinline__ R3 wl96(const uint32_t* w, uint32_t o) { R3 r; __builtin_memcpy(&r, reinterpret_cast<const uint8_t*>(w) + o, 12); return r; }
#else
__device__ __forceinline__ uint32_t wl32(const uint32_t* w, uint32_t o) { const uint32_t i = o >> 2; return __builtin_amdgcn_alignbyte(w[i + 1], w[i], o & 3u); }
__device__ __forceinline__ uint2 wl64(const uint32_t* w, uint32_t o) { const uint32_t i = o >> 2, sh = o & 3u; const uint32_t a0 = w[i], a1 = w[i + 1], a2 = w[i + 2]; return make_uint2(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh)); }
__device__ __forceinline__ uint4 wl128(const uint32_t* w, uint32_t o) {
    const uint32_t i = o >> 2, sh = o & 3u;
    const uint32_t a0 = w[i], a1 = w[i + 1], a2 = w[i + 2], a3 = w[i + 3], a4 = w[i + 4];
    return make_uint4(__builtin_amdgcn_alignbyte(a1, a0, sh), __builtin_amdgcn_alignbyte(a2, a1, sh), __builtin_amdgcn_alignbyte(a3, a2, sh), __builtin_amdgcn_alignbyte(a4, a3, sh));
}
struct R3 { uint32_t x, y, z; };
__device__ __forceinline__ R3 wl96(const uint32_t* w, uint32_t o) {      // (of z only the low byte)
    const uint32_t i = o >> 2, sh = o & 3u; const uint32_t a0 = w[i], a1 = w[i + 1], a2 = w[i + 2];
    R3 r; r.x = __builtin_amdgcn_alignbyte(a1, a0, sh); r.y = __builtin_amdgcn_alignbyte(a2, a1, sh); r.z = (a2 >> (8 * sh)) & 0xFFu; return r;
}
#endif
// Sixteen consecutive bases of a 4-bit SEQ in the LDS window as a little-endian nibble stream (base k in bits 4k .. 4k+3 of y:x),
// from base q of the SEQ section at window offset sec: the nibbles of every byte swapped (read.d:364-383: even index = high
// nibble), an odd start funnel-shifted away.
__device__ __forceinline__ uint2 nib16_w(const uint32_t* w, uint32_t sec, uint32_t q) {
    const R3 r = wl96(w, sec + (q >> 1));
    auto swp = [](uint32_t x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); };
    const uint32_t s0 = swp(r.x), s1 = swp(r.y), s2 = r.z >> 4, sh = (q & 1u) * 4u;      // (of the ninth byte only its high nibble, and only for an odd start)
    return make_uint2(__builtin_amdgcn_alignbit(s1, s0, sh), __builtin_amdgcn_alignbit(s2, s1, sh));
}
// eight nibbles (little-endian: base k in bits 4k) -> eight chars "=ACMGRSVTWYHKDBN"[code] (base.d:85): even and odd bases apart,
// four at a time through two v_perm tables and a third v_perm that picks the table by the code's bit 3, then interleaved
__device__ __forceinline__ uint2 chars8(uint32_t t) {
    auto lut = [](uint32_t x) {
        const uint32_t sel = x & 0x07070707u;
        const uint32_t lo = __builtin_amdgcn_perm(0x56535247u, 0x4D43413Du, sel);   // "=ACM" "GRSV"
        const uint32_t hi = __builtin_amdgcn_perm(0x4E42444Bu, 0x48595754u, sel);   // "TWYH" "KDBN"
        return __builtin_amdgcn_perm(hi, lo, ((x >> 1) & 0x04040404u) | 0x03020100u);
    };
    const uint32_t ev = lut(t & 0x0F0F0F0Fu), od = lut((t >> 4) & 0x0F0F0F0Fu);
    return make_uint2(__builtin_amdgcn_perm(od, ev, 0x05010400u), __builtin_amdgcn_perm(od, ev, 0x07030602u));
}
// first n bytes (nibbles) of a 64-bit word from a, the rest from b moved up behind them (1 <= n <= 15 nibbles, b's bases from its bit 0)
__device__ __forceinline__ uint2 nib_join(uint2 a, uint2 b, uint32_t n) {
    const unsigned long long A = ((unsigned long long)a.y << 32) | a.x, B = ((unsigned long long)b.y << 32) | b.x;
    const unsigned long long keep = (1ull << (4u * n)) - 1ull;
    const unsigned long long v = (A & keep) | (B << (4u * n));
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}

__device__ __forceinline__ int find_rec32(const uint32_t* off, int cnt, uint32_t b, int g) {
    g = min(max(g, 0), cnt - 1);
    if (off[g] <= b) {
        if (b < off[g + 1]) return g;
        if (g + 2 <= cnt && b < off[g + 2]) return g + 1;
    } else if (g > 0 && off[g - 1] <= b) return g - 1;
    int lo = 0, hi = cnt;
    while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (off[m] <= b) lo = m; else hi = m; }
    return lo;
}

// unpack_group for records that lie in the LDS window: soff / coff are relative to the group's first arena offsets B0 / C0,
// info[3j..] = window offsets of record j's CIGAR, SEQ and QUAL sections
template <bool SEQ, bool QUAL, bool CIG>
__device__ __forceinline__ void unpack_lds(const SdArgs& a, const uint32_t* w, int cnt, uint64_t B0, uint64_t C0, const uint32_t* soff, const uint32_t* coff, const uint16_t* info, const uint8_t* cmap) {
    const uint8_t* wb = reinterpret_cast<const uint8_t*>(w);
    if (CIG) {
        const uint32_t ctot = coff[cnt];
        const float inv = ctot ? (float)cnt / (float)ctot : 0.f;
        for (uint32_t i = threadIdx.x; i < ctot; i += SD_T) {
            const int j = find_rec32(coff, cnt, i, (int)((float)i * inv));
            a.cigar[C0 + i] = wl32(w, info[3 * j] + 4 * (i - coff[j]));
        }
    }
    if (!SEQ && !QUAL) return;
    const uint32_t tot = soff[cnt];
    if (!tot) return;
    const float inv = (float)cnt / (float)tot;
    const uint64_t b0 = B0, b1 = B0 + tot;
    const uint64_t ch0 = b0 >> 4, ch1 = (b1 + 15) >> 4;
    for (uint64_t ch = ch0 + threadIdx.x; ch < ch1; ch += SD_T) {
        const uint64_t B = ch << 4;
        const uint64_t lo = max(B, b0), hi = min(B + 16, b1);
        const uint32_t rlo = (uint32_t)(lo - b0);
        int j = cmap ? (int)cmap[ch - ch0] : find_rec32(soff, cnt, rlo, (int)((float)rlo * inv));      // (the record of the chunk's first byte: a byte per chunk, written by the records' own lanes)
        const uint32_t q0 = rlo - soff[j];
        bool done = false;
        if (lo == B && hi == B + 16) {
            const uint32_t n1 = min(16u, soff[j + 1] - rlo);
            int j2 = j + 1;
            if (n1 < 16) while (j2 < cnt && soff[j2 + 1] == soff[j2]) ++j2;
            const bool two = n1 < 16;
            const bool fits = !two || (j2 < cnt && soff[j2 + 1] - soff[j2] >= 16 - n1);
            if (fits) {
                // A chunk that straddles two records (one in nine at 150 bases a read: some lane of nearly every wave) joins the two
                // records' NIBBLES and expands once -- expanding both records to chars and merging those was 140 of the loop's 250
                // VALU instructions.  Straight-line: the second record's loads are a few LDS reads, a branch would cost as much.
                const uint32_t jb = two ? (uint32_t)j2 : (uint32_t)j, nj = two ? n1 : 8u;
                if (SEQ) {
                    uint2 nb = nib16_w(w, info[3 * j + 1], q0);
                    const uint2 rb = wl64(w, info[3 * jb + 1]);
                    auto swp = [](uint32_t x) { return ((x & 0x0F0F0F0Fu) << 4) | ((x >> 4) & 0x0F0F0F0Fu); };
                    const uint2 joined = nib_join(nb, make_uint2(swp(rb.x), swp(rb.y)), nj);
                    if (two) nb = joined;
                    const uint2 c0 = chars8(nb.x), c1 = chars8(nb.y);
                    nt_store16(a.seq + B, make_uint4(c0.x, c0.y, c1.x, c1.y));
                }
                if (QUAL) {
                    uint4 qv = wl128(w, info[3 * j + 2] + q0);
                    const uint4 q2 = wl128(w, info[3 * jb + 2] - nj);
                    if (two) qv = merge16(qv, q2, n1);
                    nt_store16(a.qual + B, qv);
                }
                done = true;
            }
        }
        if (!done) {
            for (uint64_t b = lo; b < hi; ++b) {
                const uint32_t rb = (uint32_t)(b - b0);
                while (rb >= soff[j + 1]) ++j;
                const uint32_t q = rb - soff[j];
                if (SEQ) { const uint32_t byte = wb[info[3 * j + 1] + (q >> 1)]; a.seq[b] = nt16_char((q & 1) ? (byte & 0xFu) : (byte >> 4)); }
                if (QUAL) a.qual[b] = wb[info[3 * j + 2] + q];
            }
        }
    }
}

// exclusive prefix sums of two per-thread values over the first SD_RB threads (waves 0 .. SD_RB/64-1; every thread of the
// workgroup calls) + the totals
__device__ __forceinline__ void sd_excl2(uint32_t v0, uint32_t v1, uint32_t* e0, uint32_t* e1, uint32_t* t0, uint32_t* t1, u64* ws) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t i0 = v0, i1 = v1;
    if (wv < SD_RB / 64) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o0 = __shfl_up(i0, d, 64), o1 = __shfl_up(i1, d, 64);
            if (lane >= d) { i0 += o0; i1 += o1; }
        }
        if (lane == 63) ws[wv] = ((u64)i1 << 32) | i0;
    }
    __syncthreads();
    uint32_t c0 = 0, c1 = 0, s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < SD_RB / 64; ++k) { const u64 x = ws[k]; if (k < wv) { c0 += (uint32_t)x; c1 += (uint32_t)(x >> 32); } s0 += (uint32_t)x; s1 += (uint32_t)(x >> 32); }
    *e0 = c0 + i0 - v0; *e1 = c1 + i1 - v1; *t0 = s0; *t1 = s1;
    __syncthreads();
}

// The emission of a chunk whose records' window offsets sit in roff: rows with one lane per record (fields and CIGAR
// from the LDS window, block scan for the arena offsets, coalesced stores), then the arenas with one lane per 16-byte
// chunk.  pn / pls / pnc: records, bases, CIGAR ops in front of the chunk.  A last record that runs past the window
// (last_long) is unpacked from global memory by the whole workgroup.
template <bool SEQ, bool QUAL, bool CIG, bool ROWS, bool CIG4>
__device__ __forceinline__ void sd_emit(const SdArgs& a, uint8_t* sd_lds, uint64_t wbase, uint32_t cnt, bool last_long, u64 pn, u64 pls, u64 pnc) {
    uint32_t* const w = reinterpret_cast<uint32_t*>(sd_lds);
    uint16_t* const roff = reinterpret_cast<uint16_t*>(sd_lds + L_ROFF);
    uint32_t* const soff = reinterpret_cast<uint32_t*>(sd_lds + L_SOFF);
    uint32_t* const coff = reinterpret_cast<uint32_t*>(sd_lds + L_COFF);
    uint16_t* const info = reinterpret_cast<uint16_t*>(sd_lds + L_INFO);
    u64* const misc = reinterpret_cast<u64*>(sd_lds + L_MISC);
    const int t = threadIdx.x, lane = t & 63;
    (void)lane;

    // ---- emit: rows (lane = record), then the arenas (lane = 16-byte chunk)
    const uint32_t nfast = cnt - (last_long ? 1u : 0u);
    u64 carry_seq = pls, carry_cig = pnc;
    for (uint32_t r0 = 0; r0 < nfast; r0 += SD_RB) {
        const int rc = (int)min((uint32_t)SD_RB, nfast - r0);
        const bool valid = t < rc;
        bgp_record rec{};
        uint32_t lseq = 0, ncig = 0, cig_o = 0;
        uint4 c4v = make_uint4(0, 0, 0, 0);
        if (t < 4) misc[M_TM + t] = 0;
        if (valid) {
            const uint32_t o = roff[r0 + t];
            rec.ref_id = (int32_t)wl32(w, o + 4); rec.pos = (int32_t)wl32(w, o + 8);              // read.d:907-911
            rec.bin_mq_nl = wl32(w, o + 12); rec.flag_nc = wl32(w, o + 16); rec.l_seq = (int32_t)wl32(w, o + 20);   // read.d:915-923
            rec.rec_off = wbase + o;
            const uint32_t l_name = rec.bin_mq_nl & 0xFFu, n_cig = rec.flag_nc & 0xFFFFu;
            cig_o = o + 36 + l_name;                                                                // read.d:988-990
            uint32_t covered = 0;
            if (!((rec.flag_nc >> 16) & 0x4u) && n_cig <= BGP_LONG_CIGAR) {                         // read.d:257-259; long CIGARs: k1_long_cigar afterwards
                const uint4 c4 = wl128(w, cig_o);                                                   // (a record has tags or at least SEQ/QUAL behind a short CIGAR; the window has 32 spare bytes)
                if (n_cig > 0) { c4v.x = c4.x; covered += cig_rc(c4.x) ? (c4.x >> 4) : 0u; }      // read.d:261
                if (n_cig > 1) { c4v.y = c4.y; covered += cig_rc(c4.y) ? (c4.y >> 4) : 0u; }
                if (n_cig > 2) { c4v.z = c4.z; covered += cig_rc(c4.z) ? (c4.z >> 4) : 0u; }
                if (n_cig > 3) { c4v.w = c4.w; covered += cig_rc(c4.w) ? (c4.w >> 4) : 0u; }
                for (uint32_t j = 4; j < n_cig; ++j) { const uint32_t op = wl32(w, cig_o + 4 * j); covered += cig_rc(op) ? (op >> 4) : 0u; }
            }
            rec.end_pos = (int32_t)((uint32_t)rec.pos + covered);                                   // read.d:1382
            lseq = (uint32_t)rec.l_seq; ncig = n_cig;
        }
        if (ROWS && t >= SD_RB && t - SD_RB < rc) {
            // What K4 wants to know of the round's reads (common.h: bgp_plan_row), by the waves that have no table row to make: beside the
            // row's own work it was 0.45 ms of the kernel's 13 -- two of a workgroup's eight waves hold records, the others waited for them.
            // The read's few words once more from LDS; a CIGAR of more than four ops is the planner's whatever it covers (it looks again).
            const uint32_t u = (uint32_t)t - SD_RB, o = roff[r0 + u];
            const uint32_t pos = wl32(w, o + 8), bmn = wl32(w, o + 12), fnc = wl32(w, o + 16), lq = wl32(w, o + 20);
            const uint32_t l_name = bmn & 0xFFu, n_cig = fnc & 0xFFFFu, co = o + 36 + l_name;
            const bool unmapped = ((fnc >> 16) & 0x4u) != 0;
            uint4 c4 = make_uint4(0, 0, 0, 0);
            bool cov = !unmapped && n_cig > 4;
            if (!unmapped && n_cig >= 1 && n_cig <= 4) {
                const uint4 q = wl128(w, co);
                c4.x = q.x; if (n_cig > 1) c4.y = q.y; if (n_cig > 2) c4.z = q.z; if (n_cig > 3) c4.w = q.w;
                cov = ((cig_rc(c4.x) ? c4.x >> 4 : 0u) + (cig_rc(c4.y) ? c4.y >> 4 : 0u) + (cig_rc(c4.z) ? c4.z >> 4 : 0u) + (cig_rc(c4.w) ? c4.w >> 4 : 0u)) != 0;      // basesCovered() > 0 (ops behind the CIGAR's end are 0: an M of no bases)
            }
            uint4 row;
            uint32_t cls = bgp_plan_row(pos, n_cig, lq, cov, c4, wbase + co + 4ull * n_cig, a.len, &row);
            // (K6: a CIGAR seen whole and without an N implies dna(read) as long as the read covers, md/reconstruct.d:199 -- no ops need be read for that)
            if (!unmapped && n_cig >= 1 && n_cig <= 4 && (c4.x & 15u) != 3u && (c4.y & 15u) != 3u && (c4.z & 15u) != 3u && (c4.w & 15u) != 3u) cls |= BGP_CLS_NO_SKIP;
#if BGP_K12_NT_ROWS
            nt_store16(a.prow + pn + r0 + u, row); a.rcls[pn + r0 + u] = (uint8_t)cls;
#else
            a.prow[pn + r0 + u] = row; a.rcls[pn + r0 + u] = (uint8_t)cls;
#endif
        }
        uint32_t es, ec, ts, tc;
        sd_excl2(lseq, ncig, &es, &ec, &ts, &tc, misc + M_WS);
        // which record a 16-byte chunk of the SEQ / QUAL arenas starts in: a byte per chunk, behind the records' offsets (searching the
        // offsets per chunk -- a guess and a binary search in LDS -- was a fifth of the arena loop's instructions)
        const uint32_t sh16 = (uint32_t)carry_seq & 15u, map_at = (2u * cnt + 3u) & ~3u;
        uint8_t* const cmap = (SEQ || QUAL) && map_at + ((sh16 + ts + 15u) >> 4) <= L_TAB - L_ROFF ? sd_lds + L_ROFF + map_at : nullptr;
        const uint64_t i = pn + r0 + t;
        const uint64_t tile0 = (pn + r0) / K3_TILE_RECORDS;
        if (valid) {
#if BGP_K12_NT_ROWS
            { uint4 r0v, r1v; __builtin_memcpy(&r0v, &rec, 16); __builtin_memcpy(&r1v, reinterpret_cast<const char*>(&rec) + 16, 16);
              nt_store16(reinterpret_cast<char*>(a.rec + i), r0v); nt_store16(reinterpret_cast<char*>(a.rec + i) + 16, r1v); if (CIG4) nt_store16(a.cig4 + i, c4v);
              __builtin_nontemporal_store(carry_seq + es, a.seq_off + i); __builtin_nontemporal_store(carry_cig + ec, a.cig_off + i); }
#else
            a.rec[i] = rec; a.seq_off[i] = carry_seq + es; a.cig_off[i] = carry_cig + ec; if (CIG4) a.cig4[i] = c4v;
#endif
            soff[t] = es; coff[t] = ec;
            info[3 * t] = (uint16_t)cig_o; info[3 * t + 1] = (uint16_t)(cig_o + 4 * ncig); info[3 * t + 2] = (uint16_t)(cig_o + 4 * ncig + (uint32_t)((rec.l_seq + 1) / 2));   // read.d:992-998
            if (t == rc - 1) { soff[rc] = ts; coff[rc] = tc; }
            if (cmap && lseq) {
                const uint32_t c1 = (es + lseq - 1u + sh16) >> 4;
                for (uint32_t c = es ? (es + sh16 + 15u) >> 4 : 0u; c <= c1; ++c) cmap[c] = (uint8_t)t;
            }
            if (a.tilemax) {            // K3's tile maxima: a round of 256 records touches one or two 2048-record tiles
                const K3Key kk = k3_key_of(rec.ref_id, rec.pos, rec.end_pos);
                const int sl = (i / K3_TILE_RECORDS == tile0) ? 0 : 2;
                if (kk.e) atomicMax(&misc[M_TM + sl], (u64)kk.e);
                if (kk.p) atomicMax(&misc[M_TM + sl + 1], (u64)kk.p);
            }
        }
        __syncthreads();
        if (a.tilemax && t < 4 && misc[M_TM + t]) atomicMax(&a.tilemax[2 * tile0 + t], misc[M_TM + t]);
        unpack_lds<SEQ, QUAL, CIG>(a, w, rc, carry_seq, carry_cig, soff, coff, info, cmap);
        __syncthreads();
        carry_seq += ts; carry_cig += tc;
    }
    if (last_long) {                    // the chunk's last record runs past the window: from global memory, by the whole workgroup
        uint64_t* const l_soff = reinterpret_cast<uint64_t*>(sd_lds + L_SOFF);      // [2]
        uint64_t* const l_coff = l_soff + 2;                                         // [2]
        RecInfo* const l_info = reinterpret_cast<RecInfo*>(l_soff + 4);
        if (t == 0) {
            const uint64_t off = wbase + roff[cnt - 1];
            const uint8_t* p = a.s + off;
            bgp_record rec;
            uint4 h; __builtin_memcpy(&h, p + 4, 16);
            rec.ref_id = (int32_t)h.x; rec.pos = (int32_t)h.y; rec.bin_mq_nl = h.z; rec.flag_nc = h.w;
            rec.l_seq = (int32_t)ld_u32(p + 20); rec.rec_off = off;
            const uint32_t l_name = rec.bin_mq_nl & 0xFFu, n_cig = rec.flag_nc & 0xFFFFu;
            uint32_t covered = 0;
            uint4 c4v = make_uint4(0, 0, 0, 0);
            if (!((rec.flag_nc >> 16) & 0x4u) && n_cig <= BGP_LONG_CIGAR) {
                const uint8_t* cg = p + 36 + l_name;
                for (uint32_t j = 0; j < n_cig; ++j) {
                    const uint32_t op = ld_u32(cg + 4 * j); covered += cig_rc(op) ? (op >> 4) : 0u;
                    if (j == 0) c4v.x = op; else if (j == 1) c4v.y = op; else if (j == 2) c4v.z = op; else if (j == 3) c4v.w = op;
                }
            }
            rec.end_pos = (int32_t)((uint32_t)rec.pos + covered);
            const uint64_t i = pn + cnt - 1;
            a.rec[i] = rec; a.seq_off[i] = carry_seq; a.cig_off[i] = carry_cig; if (CIG4) a.cig4[i] = c4v;
            if (ROWS) { a.prow[i] = make_uint4(0, 0, 0, 0); a.rcls[i] = (covered != 0 || (n_cig > BGP_LONG_CIGAR && !((rec.flag_nc >> 16) & 0x4u))) ? 2 : 0; }      // (a record that runs past its window: the planner's)
            if (a.tilemax) {
                const K3Key kk = k3_key_of(rec.ref_id, rec.pos, rec.end_pos);
                if (kk.e) atomicMax(&a.tilemax[2 * (i / K3_TILE_RECORDS)], (unsigned long long)kk.e);
                if (kk.p) atomicMax(&a.tilemax[2 * (i / K3_TILE_RECORDS) + 1], (unsigned long long)kk.p);
            }
            l_soff[0] = carry_seq; l_soff[1] = carry_seq + (uint64_t)(uint32_t)rec.l_seq;
            l_coff[0] = carry_cig; l_coff[1] = carry_cig + n_cig;
            RecInfo ri;
            ri.cig_ptr = off + 36 + l_name; ri.seq_ptr = ri.cig_ptr + 4ull * n_cig; ri.qual_ptr = ri.seq_ptr + (uint64_t)((rec.l_seq + 1) / 2);
            l_info[0] = ri;
        }
        __syncthreads();
        DecArgs d;
        d.s = a.s; d.stream_len = a.len; d.rec = nullptr; d.seq_off = nullptr; d.cigar_off = nullptr; d.n = 0; d.seq = a.seq; d.qual = a.qual; d.cigar = a.cigar; d.rb = 1;
        unpack_group<SEQ, QUAL, CIG>(d, 1, l_soff, l_coff, l_info);
    }
}

#ifdef BGP_EXPERIMENTS                         // the one-pass decode: exact, measured, slower (LABNOTES.md) -- not in the product build
#include "../../scratch/experiments/decode_onepass.inc"
#endif


// ---------------------------------------------------------------------------------------------- K1 + K2 through LDS windows
// The record table and the arenas in one pass over the stream, framing taken from K0: when K0's segments are SD_C bytes,
// a workgroup copies its segment (+ SD_OV) into LDS with coalesced 16-byte loads -- every stream byte is read from HBM
// once, where k1_rows and k2_unpack fetch the header lines twice and k12_fused pays for its rows with registers -- takes
// the records' offsets from the side array k0_chase left, and emits exactly as the one-pass kernel does (sd_emit).
struct LdsArgs {
    SdArgs d;                         // stream, outputs (capacities unused: the sizes are known)
    const uint64_t* rec_base; const uint64_t* seq_base; const uint64_t* cig_base; const uint32_t* side; uint32_t side_cap; int side16;
    uint64_t n_seg, n_records, sum_lseq, sum_ncig;
    const uint4* sub; uint32_t sub_n;      // a K0 segment is sub_n windows: records / bases / ops of the segment in front of each (k0_chase)
};

template <bool SEQ, bool QUAL, bool CIG, bool ROWS, bool CIG4>
__global__ __launch_bounds__(SD_T, BGP_SD_MINW) void k12_lds(LdsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t sd_lds[];
    uint16_t* const roff = reinterpret_cast<uint16_t*>(sd_lds + L_ROFF);
    const uint32_t* const w = reinterpret_cast<const uint32_t*>(sd_lds);
    const int t = threadIdx.x;
    const uint64_t k = blockIdx.x;
    if (k == 0 && t == 0) { a.d.seq_off[a.n_records] = a.sum_lseq; a.d.cig_off[a.n_records] = a.sum_ncig; }
    // window k = window h of K0's segment sg (K0 frames in segments of several windows: a guess per segment, not per window)
    const uint64_t sg = k / a.sub_n;
    const uint32_t h = (uint32_t)(k % a.sub_n);
    const uint32_t seg_cnt = (uint32_t)(a.rec_base[sg + 1] - a.rec_base[sg]);
    const uint4 p0 = h ? a.sub[sg * a.sub_n + h] : make_uint4(0, 0, 0, 0);
    const uint32_t c1 = h + 1 < a.sub_n ? a.sub[sg * a.sub_n + h + 1].x : seg_cnt;
    const uint64_t pn = a.rec_base[sg] + p0.x;
    const uint32_t cnt = c1 - p0.x;
    if (!cnt) return;                                      // (inside a record longer than a window, or behind the end of the chain)
    const uint64_t wbase = k * SD_C;
    const uint32_t wlen = (uint32_t)min((uint64_t)SD_WIN, a.d.len - wbase);
    {   // the window: every load in flight before the first store
        constexpr int NL = (SD_WIN + 32 + 16 * SD_T - 1) / (16 * SD_T);
        uint4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) { const uint32_t o = 16u * (uint32_t)(t + SD_T * i); v[i] = make_uint4(0, 0, 0, 0); if (o + 16 <= wlen) __builtin_memcpy(&v[i], a.d.s + wbase + o, 16); }
        if (a.side16) { const uint16_t* s16 = reinterpret_cast<const uint16_t*>(a.side + sg * a.side_cap); for (uint32_t j = t; j < cnt; j += SD_T) roff[j] = (uint16_t)(s16[p0.x + j] - h * SD_C); }
        else for (uint32_t j = t; j < cnt; j += SD_T) roff[j] = (uint16_t)(a.side[sg * a.side_cap + p0.x + j] - h * SD_C);
#pragma unroll
        for (int i = 0; i < NL; ++i) { const uint32_t o = 16u * (uint32_t)(t + SD_T * i); if (o < SD_WIN + 32) *reinterpret_cast<uint4*>(sd_lds + o) = v[i]; }
    }
    __syncthreads();
    if (wlen & 15u) {                                      // the last few bytes of the stream, one by one
        const uint32_t tail0 = wlen & ~15u;
        if (tail0 + t < wlen) sd_lds[tail0 + t] = a.d.s[wbase + tail0 + t];
        __syncthreads();
    }
    const bool last_long = (uint64_t)roff[cnt - 1] + 4 + (uint64_t)wl32(w, roff[cnt - 1]) > wlen;
    sd_emit<SEQ, QUAL, CIG, ROWS, CIG4>(a.d, sd_lds, wbase, cnt, last_long, pn, a.seq_base[sg] + p0.y, a.cig_base[sg] + p0.z);
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
    // 256 records a workgroup -- unless that leaves the device short of workgroups (200 k reads of 30 kb: 780 of them for 2048
    // slots, 9.5 ms): then as few records each as it takes to have ~16 k workgroups
    a.rb = (uint32_t)std::min<uint64_t>(RB, std::max<uint64_t>(1, n / 16384));
    a.rec_fix = nullptr;
    if (c->long_cigar_deferred && (what & BGP_FETCH_CIGAR)) { a.rec_fix = c->d_rec.as<bgp_record>(); c->long_cigar_deferred = false; }
    const unsigned blocks = (unsigned)((n + a.rb - 1) / a.rb);
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
    BGP_TRY(c->d_cig4.reserve((n + 1) * 16)); c->cig4_ok = true; c->rows_ok = false; c->lds_table = false;
    if (what & BGP_FETCH_SEQ) BGP_TRY(c->d_seq.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_QUAL) BGP_TRY(c->d_qual.reserve(c->sum_lseq + 16));
    if (what & BGP_FETCH_CIGAR) BGP_TRY(c->d_cigar.reserve(c->sum_ncig * 4 + 16));
    c->decoded_what |= what;
    if (c->stream_len < 4) return BGP_OK;
    FusedArgs a;
    a.d.s = c->d_stream; a.d.stream_len = c->stream_len; a.d.rec = nullptr; a.d.seq_off = nullptr; a.d.cigar_off = nullptr; a.d.n = n; a.d.rb = RB;
    a.d.seq = c->d_seq.as<uint8_t>(); a.d.qual = c->d_qual.as<uint8_t>(); a.d.cigar = c->d_cigar.as<uint32_t>();
    a.entry = c->d_seg_entry.as<uint64_t>(); a.rec_base = c->d_seg_rec_base.as<uint64_t>(); a.seq_base = c->d_seg_seq_base.as<uint64_t>(); a.cig_base = c->d_seg_cig_base.as<uint64_t>();
    a.side = c->d_seg_side.as<uint32_t>(); a.side_cap = c->side_cap; a.side16 = c->side16 ? 1 : 0; a.seg_bytes = c->seg_bytes; a.n_seg = c->n_seg; a.sum_lseq = c->sum_lseq; a.sum_ncig = c->sum_ncig;
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
    uint64_t alg = n * (36 + 32 + 16) + (c->defer_long_cigar && c->long_cigar ? std::min<uint64_t>(c->sum_ncig, 4 * n) : c->sum_ncig) * 4;    // what k1_emit accounts for (the long CIGARs only when they are summed here, not by the unpack that follows) ...
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

// K0 + K1 (+ K2 for `what`) in one pass over the stream: framing, record table, arenas (k_stream_decode above).
// Returns 1 if the pass gave up (a look-back that never resolved: not expected; the caller then takes the three-kernel path).
#ifndef BGP_EXPERIMENTS
int k012_stream_decode(bgp_ctx*, uint32_t) { return 1; }     // not built: the caller takes K0 + K1 + K2
#else
#include "../../scratch/experiments/decode_onepass_host.inc"
#endif

// K1 + K2 through LDS windows (k12_lds): when K0 framed the stream in SD_C-byte segments and every section is wanted
#ifndef BGP_LDS_WIN_PER_SEG
#define BGP_LDS_WIN_PER_SEG 2          // windows of k12_lds per framing segment of K0: a guess per 64 KiB instead of per 32 (cfg3 k0_guess 1.25 -> 0.64 ms, the chase 4.67 -> 4.86; with four windows 0.34 and 5.24: the chains of the chase get long)
#endif
static_assert((SD_C & (SD_C - 1)) == 0, "window = 1 << shift");
bool k12_lds_applies(const bgp_ctx* c, uint32_t what) { return c->table_pending && c->seg_windows && c->seg_bytes == SD_C * c->seg_windows && (what & BGP_FETCH_ALL) == BGP_FETCH_ALL && c->n_seg * c->seg_windows < 0x7FFFFFFFull; }
// (a stream of up to a GB or two has too few segments to fill the device with chases -- one lane each, ~230 dependent hops at 64 KiB:
// 0.17 ms whatever the size below ~100 MB --: one window a segment then, twice the lanes and half the hops)
uint32_t k12_lds_segment_bytes(uint64_t stream_len) { return SD_C * (stream_len < (2048ull << 20) ? 1u : (uint32_t)BGP_LDS_WIN_PER_SEG); }
// The side array of first CIGAR ops (16 bytes a record) when k12_lds left it out because it wrote plan rows: made from the CIGAR arena -- dense,
// where the records' CIGARs lie 290 bytes apart in the stream -- the first time somebody wants it (K5's entries, K6's full rows).
namespace {
__global__ __launch_bounds__(256) void k_cig4_from_arena(const bgp_record* rec, const uint64_t* cig_off, const uint32_t* cigar, uint64_t n, uint4* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t n_cig = rec[i].flag_nc & 0xFFFFu;
    const uint32_t* q = cigar + cig_off[i];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);                      // (ops behind the CIGAR's end are 0, as K1 leaves them)
    if (n_cig > 0) v.x = q[0];
    if (n_cig > 1) v.y = q[1];
    if (n_cig > 2) v.z = q[2];
    if (n_cig > 3) v.w = q[3];
    out[i] = v;
}
}  // namespace
int ensure_cig4(bgp_ctx* c) {
    if (c->cig4_ok || !c->lds_table || !c->n_records) return BGP_OK;      // there already; or no table of k12_lds's with its arenas: the consumers read the stream
    const uint64_t n = c->n_records;
    BGP_TRY(c->d_cig4.reserve((n + 1) * 16));
    TimedRegion t(c, "k_cig4_from_arena", n * 44);
    hipLaunchKernelGGL(k_cig4_from_arena, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_rec.as<bgp_record>(), c->d_cigar_off.as<uint64_t>(), c->d_cigar.as<uint32_t>(), n, c->d_cig4.as<uint4>());
    BGP_HIP(hipGetLastError());
    c->cig4_ok = true;
    return BGP_OK;
}

uint32_t k12_lds_window_shift() { uint32_t s = 0; while ((1u << s) < SD_C) ++s; return s; }
int k12_lds_decode(bgp_ctx* c, uint32_t what) {
    const uint64_t n = c->n_records;
    c->table_pending = false;
    BGP_TRY(c->d_rec.reserve((n + 1) * sizeof(bgp_record)));
    BGP_TRY(c->d_seq_off.reserve((n + 1) * 8)); BGP_TRY(c->d_cigar_off.reserve((n + 1) * 8));
    static const bool rows_env = !(getenv("BGP_K4_ROWS") && atoi(getenv("BGP_K4_ROWS")) == 0);      // (0: K4 plans every read itself, as it does behind the other decode routes)
    const bool decode_only = (what & BGP_DECODE_ONLY) != 0;      // the caller's word that no pileup follows: nothing is written for K4 (a pileup asked for anyway has the side array made from the arena)
    const bool want_rows = rows_env && !decode_only && !c->odd_reads;      // (a stream with a read of more than four ops or 255 bases is planned read by read, its odd reads walked from checkpoints: rows would be written for nobody -- framing has seen every header)
    c->rows_ok = false; c->lds_table = false;
    // With rows the first-ops side array (16 bytes a record: what K4's planner, K5 and K6 read instead of the stream) is not written here: the
    // planner only sees the few reads a row cannot hold, K6 takes the CIGAR-implied length from the class byte, and K5 has it made from the
    // CIGAR arena when entries are first asked for (ensure_cig4) -- 1.6 GB of stores the step itself has no reader for.
    static const bool cig4_with_rows = getenv("BGP_K12_CIG4") && atoi(getenv("BGP_K12_CIG4")) != 0;
    const bool want_cig4 = (!want_rows && !decode_only) || cig4_with_rows;
    c->cig4_ok = false;
    if (want_cig4) { BGP_TRY(c->d_cig4.reserve((n + 1) * 16)); c->cig4_ok = true; }
    if (want_rows) { BGP_TRY(c->d_prow.reserve((n + 1) * 16)); BGP_TRY(c->d_rcls.reserve(n + 16)); }
    BGP_TRY(c->d_seq.reserve(c->sum_lseq + 16)); BGP_TRY(c->d_qual.reserve(c->sum_lseq + 16)); BGP_TRY(c->d_cigar.reserve(c->sum_ncig * 4 + 16));
    c->decoded_what |= BGP_FETCH_ALL;
    if (c->stream_len < 4) return BGP_OK;
    LdsArgs a;
    a.d.s = c->d_stream; a.d.len = c->stream_len; a.d.n_chunks = c->n_seg; a.d.ref_bound = 0; a.d.tolerant = 0; a.d.hint = nullptr; a.d.desc = nullptr; a.d.ctl = nullptr; a.d.status = nullptr;
    a.d.rec = c->d_rec.as<bgp_record>(); a.d.seq_off = c->d_seq_off.as<uint64_t>(); a.d.cig_off = c->d_cigar_off.as<uint64_t>(); a.d.cig4 = want_cig4 ? c->d_cig4.as<uint4>() : nullptr;
    a.d.seq = c->d_seq.as<uint8_t>(); a.d.qual = c->d_qual.as<uint8_t>(); a.d.cigar = c->d_cigar.as<uint32_t>();
    a.d.cap_n = ~0ull; a.d.cap_ls = ~0ull; a.d.cap_nc = ~0ull;
    a.d.prow = want_rows ? c->d_prow.as<uint4>() : nullptr; a.d.rcls = want_rows ? c->d_rcls.as<uint8_t>() : nullptr; c->rows_ok = want_rows; c->lds_table = true;
    a.d.tilemax = nullptr; c->tilemax_ok = false;
    if (!c->long_cigar) {             // K3's tile maxima on the way (see k12_fused_decode)
        const uint64_t nt = (n + K3_TILE_RECORDS - 1) / K3_TILE_RECORDS;
        BGP_TRY(c->d_tilemax.reserve((nt + 2) * 16));
        BGP_HIP(hipMemsetAsync(c->d_tilemax.p, 0, (nt + 2) * 16, c->stream));
        a.d.tilemax = (unsigned long long*)c->d_tilemax.p; c->tilemax_ok = true;
    }
    a.rec_base = c->d_seg_rec_base.as<uint64_t>(); a.seq_base = c->d_seg_seq_base.as<uint64_t>(); a.cig_base = c->d_seg_cig_base.as<uint64_t>();
    a.side = c->d_seg_side.as<uint32_t>(); a.side_cap = c->side_cap; a.side16 = c->side16 ? 1 : 0; a.n_seg = c->n_seg; a.n_records = n; a.sum_lseq = c->sum_lseq; a.sum_ncig = c->sum_ncig;
    a.sub = c->d_seg_sub.as<uint4>(); a.sub_n = c->seg_windows;
    const uint64_t n_win = std::min<uint64_t>(c->n_seg * c->seg_windows, (c->stream_len + SD_C - 1) / SD_C);
    {
        TimedRegion t(c, "k12_lds", c->stream_len + n * 32 + 2 * c->sum_lseq + 4 * c->sum_ncig);      // SURVEY 8(d) without the framing: the record in, row + seq + qual + cigar out
        // what K4 is left (plan rows, the first-ops side array) is compiled in or out: the rows' block behind a run-time test cost every variant
        // registers -- 64 VGPRs with a spill to scratch and 56 spilled SGPRs against 62 / 0 / 29 without it (round 5's lost 0.7 ms)
#ifndef BGP_K12_S
#define BGP_K12_S true       // (timing experiments, wrong arenas: false leaves a section's work and stores out)
#define BGP_K12_Q true
#define BGP_K12_C true
#endif
#define LAUNCH(r_, c_) hipLaunchKernelGGL((k12_lds<BGP_K12_S, BGP_K12_Q, BGP_K12_C, r_, c_>), dim3((unsigned)n_win), dim3(SD_T), L_TOTAL, c->stream, a)
        if (want_rows && want_cig4) LAUNCH(true, true);
        else if (want_rows) LAUNCH(true, false);
        else if (want_cig4) LAUNCH(false, true);
        else LAUNCH(false, false);
#undef LAUNCH
        BGP_TRY(k1_finish_long_cigars(c));
    }
    BGP_HIP(hipGetLastError());
    (void)what;
    return BGP_OK;
}
