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
    uint2 w; __builtin_memcpy(&w, p, 8);
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
                if (SEQ) *reinterpret_cast<uint4*>(a.seq + B) = sv;
                if (QUAL) *reinterpret_cast<uint4*>(a.qual + B) = qv;
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
