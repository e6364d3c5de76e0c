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
    const uint8_t* s; const bgp_record* rec; const uint64_t* seq_off; const uint64_t* cigar_off; uint64_t n;
    uint8_t* seq; uint8_t* qual; uint32_t* cigar;
};

__device__ __forceinline__ uint8_t nt16_char(uint32_t code) {      // base.d:85
    // "=ACMGRSV" "TWYHKDBN" as little-endian u64 constants
    const uint64_t lo = 0x565352474D43413Dull, hi = 0x4E42444B48595754ull;
    return (uint8_t)(((code & 8u) ? hi : lo) >> ((code & 7u) * 8));
}

struct RecInfo { uint64_t cig_ptr, seq_ptr, qual_ptr; };

__device__ __forceinline__ int find_rec(const uint64_t* off /*LDS, cnt+1*/, int cnt, uint64_t b) {
    int lo = 0, hi = cnt;             // largest j with off[j] <= b  (off[cnt] > b guaranteed by caller)
    while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (off[m] <= b) lo = m; else hi = m;
    }
    return lo;
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
        for (uint64_t i = c0 + threadIdx.x; i < c1; i += blockDim.x) {
            const int j = find_rec(s_cig_off, cnt, i);
            a.cigar[i] = ld_u32(a.s + s_info[j].cig_ptr + 4 * (i - s_cig_off[j]));
        }
    }
    if (!SEQ && !QUAL) return;

    const uint64_t b0 = s_seq_off[0], b1 = s_seq_off[cnt];
    if (b1 == b0) return;
    const uint64_t ch0 = b0 >> 4, ch1 = (b1 + 15) >> 4;
    for (uint64_t ch = ch0 + threadIdx.x; ch < ch1; ch += blockDim.x) {
        const uint64_t B = ch << 4;
        const uint64_t lo = max(B, b0), hi = min(B + 16, b1);     // bytes of this chunk this workgroup owns
        int j = find_rec(s_seq_off, cnt, lo);
        const uint64_t q0 = lo - s_seq_off[j];
        if (lo == B && hi == B + 16 && B + 16 <= s_seq_off[j + 1]) {
            // fast path: the whole chunk lies inside record j
            if (SEQ) {
                const uint8_t* src = a.s + s_info[j].seq_ptr + (q0 >> 1);
                uint64_t w; __builtin_memcpy(&w, src, 8);
                const uint32_t extra = src[8];                     // in-record: QUAL follows SEQ
                uint32_t out[4];
                const uint32_t odd = (uint32_t)q0 & 1u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t t = odd + i;                    // nibble index from the first loaded byte
                    const uint32_t byte = (t >> 1) < 8 ? (uint32_t)(w >> ((t >> 1) * 8)) & 0xFFu : extra;
                    const uint32_t code = (t & 1u) ? (byte & 0xFu) : (byte >> 4);   // read.d:368-373
                    if ((i & 3) == 0) out[i >> 2] = 0;
                    out[i >> 2] |= (uint32_t)nt16_char(code) << ((i & 3) * 8);
                }
                *reinterpret_cast<uint4*>(a.seq + B) = make_uint4(out[0], out[1], out[2], out[3]);
            }
            if (QUAL) {
                uint4 v; __builtin_memcpy(&v, a.s + s_info[j].qual_ptr + q0, 16);
                *reinterpret_cast<uint4*>(a.qual + B) = v;
            }
        } else {
            // boundary chunk: walk bytes, advancing the record as offsets are crossed
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
    a.s = c->d_stream; a.rec = c->d_rec.as<bgp_record>(); a.seq_off = c->d_seq_off.as<uint64_t>(); a.cigar_off = c->d_cigar_off.as<uint64_t>();
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
