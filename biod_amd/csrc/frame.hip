// frame.hip -- kernels K0 (record framing) and K1 (fixed fields, offsets, end_position).
//
// Reference semantics: BamReadRange.readNext (bio/std/hts/bam/readrange.d:118-173) -- a serial
// pointer chase `next = off + 4 + block_size` -- and the BamRead field accessors
// (bio/std/hts/bam/read.d:907-1003), basesCovered (read.d:255-262), EagerBamRead (read.d:1380-1383).
//
// MI355X design.  The chase is the only serial dependency of the whole path, so it is cut into
// fixed-size byte segments and run speculatively:
//   k0_guess   one wave per segment scans for the first *plausible* record header (64 candidate
//              offsets per step, ballot picks the first whose successor header is plausible too).
//   k0_chase   one lane per segment chases length prefixes from the segment's entry until it
//              leaves the segment; emits (chain end, #records, sum l_seq, sum n_cigar).
//   k0_verify  entry[k] must equal chain_end[k-1]; mismatches are corrected and re-chased until
//              no segment is dirty.  At the fixpoint every entry is, by induction from entry[0] = 0,
//              the true record boundary -- the guess only ever costs time, never correctness.
// Exclusive scans of the per-segment counts give every segment its base record index and arena
// offsets; k1_emit re-chases (L2-warm) and writes the 32-byte record table.
#include "common.h"
#include <cstdio>

namespace {

constexpr uint64_t NONE = ~0ull;        // entry/end not known (speculation found nothing)
constexpr uint64_t STOP = ~0ull - 1;    // chain is over (error upstream); segment holds no records
constexpr uint32_t LONG_CIGAR = BGP_LONG_CIGAR;

struct FrameArgs {
    const uint8_t* s; uint64_t len; uint32_t seg_bytes; uint64_t n_seg; int32_t ref_bound;
    uint64_t* entry; uint64_t* fix; uint64_t* end; uint64_t* cnt; uint64_t* lseq; uint64_t* ncig; uint32_t* dirty; uint32_t* seg_err;
    uint32_t* status; /* [0] status bits, [1] dirty counter */
    uint32_t* side; uint32_t side_cap;   // side[k * side_cap + j] = offset of segment k's j-th record relative to the segment start
    int side16;                          // ... as 16-bit values (segments of at most 64 KiB), two to a word of the same array
    const uint64_t* seg_hint;            // per segment: the smallest offset inside it where the caller says a record starts (NONE: no word), or null
    int tolerant_tail;                   // batched staging: the data may end inside a record -- the chain stops in front of it instead of ETRUNC
    uint32_t* list; uint32_t list_cap;   // k0_verify lists the segments it sends round again (k0_rechase takes them from there)
    uint4* sub; uint32_t sub_n, sub_shift;   // k12_lds: a segment is sub_n windows of 1 << sub_shift bytes; sub[k * sub_n + w] = records, bases, CIGAR ops of segment k in front of window w (null: not wanted)
};

// one wave per segment
__global__ __launch_bounds__(256) void k0_guess(FrameArgs a) {
    const uint64_t k = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= a.n_seg) return;
    if (k == 0) { if (lane == 0) a.entry[0] = 0; return; }
    const uint64_t beg = k * a.seg_bytes;
    const uint64_t lim = min(beg + a.seg_bytes, a.len);
    uint64_t found = NONE;
    if (a.seg_hint) {                                   // the caller's hint (bgp_push_inflated: first_record_uoffset): tried first, by the same test as any candidate
        const uint64_t h = a.seg_hint[k];
        uint32_t bs = 0, dummy;
        if (h != NONE && h >= beg && h < lim && plausible(a.s, a.len, h, a.ref_bound, &bs)) {
            const uint64_t nxt = h + 4 + bs;
            if (a.len - nxt < 4 || plausible(a.s, a.len, nxt, a.ref_bound, &dummy)) found = h;
        }
    }
    // Second pass, only by a wave whose first found nothing, with the widest reference bound and without the printable-name test: a
    // caller's wrong bgp_set_n_ref, or a file whose read names are not printable, then costs this pass instead of one lane chasing the
    // whole stream serially (measured on cfg3's model with n_ref set to 1 where there are 8: INTEGRATION.md "Limits").  A segment that
    // holds no record start is scanned twice.  (A launch of its own until round 4: 110 k workgroups that looked at one word and left.)
    for (int pass = 0; pass < 2 && found == NONE; ++pass) {
        const bool names = pass == 0;
        const int32_t ref_bound = pass == 0 ? a.ref_bound : (1 << 24);
        // A wave's time is its chain of round trips: the next step's eight bytes a lane are asked for before this step's are looked at,
        // and a candidate that passes them asks for the rest of its header, for its successor's eight bytes and for the rest of the
        // successor's header TOGETHER (the successor's offset is known from block_size) -- two round trips a segment's true header where
        // there were four, and the steps in front of it no longer one each.  (256 candidates a step, four to a lane as in k0_guess_wide:
        // 0.65 against 0.52 ms -- the start is half a record in, and three quarters of the wider step's loads are for nothing.)
        auto ld01 = [&](uint64_t o, uint32_t (&h01)[2]) { h01[0] = h01[1] = 0u; if (o < lim && o + 36 <= a.len) __builtin_memcpy(h01, a.s + o, 8); };
        uint32_t cur[2];
        ld01(beg + lane, cur);
        for (uint64_t base = beg; base < lim && found == NONE; base += 64) {
            uint32_t nx01[2];
            ld01(base + 64 + lane, nx01);
            const uint64_t o = base + lane;
            const bool q = o < lim && o + 36 <= a.len && plausible_quick(cur, a.len, o, ref_bound);
            if (__ballot(q)) {
                bool ok = false;
                if (q) {
                    const uint32_t bs = cur[0];
                    const uint64_t nxt = o + 4 + bs;
                    uint32_t hc[10], hs01[2] = {0u, 0u}, hs[10];
                    bool wc, ws = false;
                    plausible_load_rest(a.s, a.len, o, hc, &wc);
                    const bool s_in = nxt + 36 <= a.len;
                    if (s_in) { __builtin_memcpy(hs01, a.s + nxt, 8); plausible_load_rest(a.s, a.len, nxt, hs, &ws); }
                    else { for (int k = 0; k < 10; ++k) hs[k] = 0u; }
                    ok = plausible_rest(a.s, a.len, o, bs, hc, wc, ref_bound, names)
                         && (a.len - nxt < 4 || (s_in && plausible_quick(hs01, a.len, nxt, ref_bound) && plausible_rest(a.s, a.len, nxt, hs01[0], hs, ws, ref_bound, names)));
                }
                const unsigned long long mm = __ballot(ok);
                if (mm) found = base + (uint64_t)(__ffsll((long long)mm) - 1);
            }
            cur[0] = nx01[0]; cur[1] = nx01[1];
        }
    }
    if (lane == 0) a.entry[k] = found;
}

// The same guess with a whole workgroup per segment, 256 candidates a step: for streams of few, large segments (long
// reads: 1 MiB segments, the first record start tens of KB in), where one wave per segment leaves the device idle and
// walks ~500 dependent steps (cfg5: 2.5 ms).
__global__ __launch_bounds__(256) void k0_guess_wide(FrameArgs a) {
    __shared__ unsigned long long s_found;
    const uint64_t k = blockIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (k >= a.n_seg) return;
    if (k == 0) { if (threadIdx.x == 0) a.entry[0] = 0; return; }
    const uint64_t beg = k * a.seg_bytes;
    const uint64_t lim = min(beg + a.seg_bytes, a.len);
    if (threadIdx.x == 0) {
        uint64_t f = NONE;
        if (a.seg_hint) {
            const uint64_t h = a.seg_hint[k];
            uint32_t bs = 0, dummy;
            if (h != NONE && h >= beg && h < lim && plausible(a.s, a.len, h, a.ref_bound, &bs)) {
                const uint64_t nxt = h + 4 + bs;
                if (a.len - nxt < 4 || plausible(a.s, a.len, nxt, a.ref_bound, &dummy)) f = h;
            }
        }
        s_found = f;
    }
    __syncthreads();
    uint64_t found = s_found;
    // 1024 candidates a step, four to a thread: block_size and refID of all four are requested together (what turns most candidates
    // down, as in plausible()); a lane tests in full what passes them -- every lane its own, side by side: inside a long read's CIGAR
    // every aligned offset passes the first two tests -- and only the successor test of a full pass is the wave's.  With 256 a step
    // the first record start of a 4 MiB segment of long reads -- tens of KB in -- was ~100 dependent steps (cfg5: 0.66 ms).
    for (int pass = 0; pass < 2 && found == NONE; ++pass) {      // (the second pass: see k0_guess)
    const bool names = pass == 0;
    const int32_t ref_bound = pass == 0 ? a.ref_bound : (1 << 24);
    for (uint64_t base = beg; found == NONE && base < lim; base += 1024) {
        uint64_t mine = NONE;
        uint32_t pre[4][2], bs4[4]; bool ok4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t o = base + 256u * u + 64u * wv + lane;
            pre[u][0] = pre[u][1] = 0u;
            if (o < lim && o + 36 <= a.len) __builtin_memcpy(pre[u], a.s + o, 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t o = base + 256u * u + 64u * wv + lane;
            const uint32_t bs0 = pre[u][0];
            const bool quick = o < lim && o + 36 <= a.len && bs0 >= 33u && bs0 <= (1u << 30) && (uint64_t)bs0 <= a.len - o - 4 && (int32_t)pre[u][1] >= -1 && (int32_t)pre[u][1] < ref_bound;
            bs4[u] = 0;
            ok4[u] = quick && plausible(a.s, a.len, o, ref_bound, &bs4[u], names);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned long long m = mine == NONE ? __ballot(ok4[u]) : 0ull;
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                const uint32_t bsl = __shfl(bs4[u], l, 64);
                const uint64_t ol = base + 256u * u + 64u * wv + l, nxt = ol + 4 + bsl;
                uint32_t dummy;
                if (a.len - nxt < 4 || plausible(a.s, a.len, nxt, ref_bound, &dummy, names)) { mine = ol; break; }
                m &= m - 1;
            }
        }
        if (lane == 0 && mine != NONE) atomicMin(&s_found, (unsigned long long)mine);
        __syncthreads();
        found = s_found;
        __syncthreads();
    }
    }
    if (threadIdx.x == 0) a.entry[k] = found;
}

constexpr int SIDE_GROUP = 16;          // offsets a lane of k0_chase collects before it stores them (one 64-byte line)

// where the chase reads the stream from: global memory, or the segment's bytes staged in LDS (k0_rechase) with global memory behind them
struct RdGlobal {
    const uint8_t* s;
    __device__ __forceinline__ uint32_t u32(uint64_t off) const { return ld_u32(s + off); }
    __device__ __forceinline__ void f3(uint64_t off, uint32_t* o) const { __builtin_memcpy(o, s + off, 12); }
};
struct RdLds {
    const uint8_t* s; const uint8_t* w; uint64_t w0, w1;      // window = stream bytes [w0, w1)
    __device__ __forceinline__ uint32_t u32(uint64_t off) const { uint32_t v; if (off >= w0 && off + 4 <= w1) __builtin_memcpy(&v, w + (off - w0), 4); else v = ld_u32(s + off); return v; }
    __device__ __forceinline__ void f3(uint64_t off, uint32_t* o) const { if (off >= w0 && off + 12 <= w1) __builtin_memcpy(o, w + (off - w0), 12); else __builtin_memcpy(o, s + off, 12); }
};

// The chase of ONE segment k from `off` (readrange.d:118-173): every record start inside the segment, its offset to the side array,
// the segment's counts and where the chain leaves it.  Returns true when the chain goes on into a following segment the guesser
// found nothing in (the caller chases that one right away from *off_io).
// SINK: how the records' offsets (relative to the segment) reach the side array.
//   0  32-bit offsets, sixteen at a time from registers (any segment size): every active lane of the wave is at the same record
//      count, so the buffer's indices are static, and a lane's stores are whole 64-byte groups.  One 4-byte store per record --
//      10^8 partial-sector writes the L2 cannot merge, ~20 us apart per lane -- cost this kernel 1.6 of its 4.6 ms (scratch/chase_probe.hip).
//   1  16-bit offsets (segments of at most 64 KiB: the LDS-window decode's), thirty-two at a time through a ring in LDS
//      (ring[entry][thread]: the lanes of a wave write neighbouring halfwords): half the bytes and half the stores of sink 0 -- by
//      the probe's arithmetic the stores were 0.7 of the chase's 3.3 ms, 0.3 of it per store and 0.4 per byte.
//   2  16-bit offsets stored one by one (k0_rechase: one thread, a handful of segments)
constexpr int RING_N = 32;
template <class RD, int SINK> __device__ __forceinline__ bool chase_one(const FrameArgs& a, uint64_t k, uint64_t* off_io, const RD& rd, uint16_t* ring = nullptr) {
    uint64_t off = *off_io;
    const uint64_t seg_end = min((k + 1) * (uint64_t)a.seg_bytes, a.len);
    uint64_t cnt = 0, ls = 0, nc = 0;
    uint32_t err = 0;
    bool long_seen = false, odd_seen = false;
    uint32_t wcur = 0;
    const uint64_t seg0 = k * (uint64_t)a.seg_bytes;
    bool live = off < seg_end;
    // one hop: true when a record starts at `off` and passes the reference's tests -- its offset in *v, the sums and `off` moved on;
    // `live` says whether the chain goes on inside the segment
    auto hop = [&](uint32_t* v) -> bool {
        if (a.sub) {                                                     // the windows this record's start has passed: what lay in front of them
            const uint32_t w = off >= seg0 ? (uint32_t)((off - seg0) >> a.sub_shift) : 0u;      // (a batch's chain may stand in front of the segment: a record it still waits for)
            while (wcur < w && wcur + 1 < a.sub_n) { ++wcur; a.sub[k * a.sub_n + wcur] = make_uint4((uint32_t)cnt, (uint32_t)ls, (uint32_t)nc, 0u); }
        }
        live = false;
        if (a.len - off < 4) { if (!a.tolerant_tail) off = a.len; return false; }     // readrange.d:139-149: short prefix ends the range (a batch keeps it for the next one)
        // block_size and the three header words the chase sums (bin_mq_nl, flag_nc, l_seq: bytes 12 .. 23) are requested TOGETHER.
        // Asked for only once block_size had come back and passed its tests, the second load found the line evicted again --
        // 442 k lanes keep 56 MB of lines in flight, the L2s hold 32 -- and fetched it a second time: 1.8 lines a record by
        // FETCH_SIZE where the two loads touch 1.18 (profiles/r03_pmc_chase_probe.txt).
        uint32_t f3[3] = {0u, 0u, 0u};
        const bool has_f3 = a.len - off >= 24;
        const int32_t bs = (int32_t)rd.u32(off);
        if (has_f3) rd.f3(off + 12, f3);
        asm volatile("; header words settled" : "+v"(f3[0]), "+v"(f3[1]), "+v"(f3[2]));      // (keeps the load here: sunk into the branches that use it, it would be issued late again)
        if (bs < 0) { err = ST_EINVAL; return false; }
        if ((uint64_t)bs > a.len - off - 4) {                        // readrange.d:169; in a batch the rest of the record may still be to come --
            if (!a.tolerant_tail) err = ST_ETRUNC;                    // but only a record that can exist is waited for: a corrupt block_size in the
            else if (bs < 32 || (uint32_t)bs > (1u << 30)) err = ST_EINVAL;   // middle of a stream would otherwise be carried from batch to batch, growing,
            else if (has_f3) {                                        // and surface only at bgp_batch_end (the bound plausible() uses; the header
                const int32_t lseq = (int32_t)f3[2];                      // fields that are there pass the chase's own checks)
                const uint32_t l_name = f3[0] & 0xFFu, n_cig = f3[1] & 0xFFFFu;
                if (lseq < 0 || l_name == 0 || 32ull + l_name + 4ull * n_cig + (uint64_t)((lseq + 1) / 2) + (uint64_t)lseq > (uint64_t)bs) err = ST_EINVAL;
            }
            return false;
        }
        if (bs < 32) { err = ST_EINVAL; return false; }
        const uint32_t bmn = f3[0], fnc = f3[1];                     // (bs >= 32 and inside the data: the three words were there)
        const int32_t lseq = (int32_t)f3[2];
        const uint32_t l_name = bmn & 0xFFu, n_cig = fnc & 0xFFFFu;
        if (lseq < 0 || l_name == 0 || 32ull + l_name + 4ull * n_cig + (uint64_t)((lseq + 1) / 2) + (uint64_t)lseq > (uint64_t)bs) { err = ST_EINVAL; return false; }
        *v = (uint32_t)(off - seg0);
        ++cnt; ls += (uint64_t)lseq; nc += n_cig;
        long_seen |= n_cig > LONG_CIGAR; odd_seen |= n_cig > BGP_PLAN_MAX_OPS || lseq > 255;
        off += 4 + (uint64_t)bs;
        live = off < seg_end;
        return true;
    };
    if (SINK == 0) {
        uint32_t* dst = a.side + k * a.side_cap;
        while (live) {
            uint32_t buf[SIDE_GROUP];
#pragma unroll
            for (int e = 0; e < SIDE_GROUP; ++e) {
                buf[e] = 0;
                if (live) { uint32_t v = 0; if (hop(&v)) buf[e] = v; }
            }
            if (dst + SIDE_GROUP <= a.side + (k + 1) * a.side_cap) {                 // (always: a record is at least 36 bytes and side_cap is sized for that)
#pragma unroll
                for (int q = 0; q < SIDE_GROUP / 4; ++q) reinterpret_cast<uint4*>(dst)[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
            }
            dst += SIDE_GROUP;
        }
    } else if (SINK == 1) {
        uint16_t* const dst16 = reinterpret_cast<uint16_t*>(a.side + k * a.side_cap);
        const uint32_t cap16 = 2u * a.side_cap;
        uint16_t* const mine = ring + threadIdx.x;
        uint32_t g = 0, flushed = 0;
        auto flush = [&]() {
            if (flushed + RING_N <= cap16) {
#pragma unroll
                for (int q = 0; q < RING_N / 8; ++q) {
                    uint32_t wv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const uint32_t lo = (uint32_t)(8 * q + 2 * e) < g ? mine[(8 * q + 2 * e) * 256] : 0u, hi = (uint32_t)(8 * q + 2 * e + 1) < g ? mine[(8 * q + 2 * e + 1) * 256] : 0u; wv[e] = lo | (hi << 16); }
                    reinterpret_cast<uint4*>(dst16 + flushed)[q] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                }
            }
            flushed += RING_N; g = 0;
        };
        while (live) {
            uint32_t v = 0;
            if (hop(&v)) { mine[g * 256] = (uint16_t)v; if (++g == RING_N) flush(); }
        }
        if (g) flush();
    } else {
        uint16_t* const dst16 = reinterpret_cast<uint16_t*>(a.side + k * a.side_cap);
        const uint32_t cap16 = 2u * a.side_cap;
        uint32_t n = 0;
        while (live) { uint32_t v = 0; if (hop(&v)) { if (n < cap16) dst16[n] = (uint16_t)v; ++n; } }
    }
    if (a.sub) while (wcur + 1 < a.sub_n) { ++wcur; a.sub[k * a.sub_n + wcur] = make_uint4((uint32_t)cnt, (uint32_t)ls, (uint32_t)nc, 0u); }
    a.end[k] = err ? STOP : off;
    // per segment, not a sticky global bit: a chase from a wrong guess reads garbage and must not leave a trace
    a.cnt[k] = cnt; a.lseq[k] = ls; a.ncig[k] = nc; a.seg_err[k] = err | (long_seen ? ST_LONGCIG : 0u) | (odd_seen ? ST_ODDREAD : 0u);
    *off_io = off;
    // A following segment the guesser found nothing in (inside a record longer than a segment, or bytes that
    // look like no header) is chased right away from where this chain stands -- otherwise every such segment
    // would cost k0_verify one more round, host round trip included.  Its own lane wrote nothing (see k0_chase);
    // should it read the entry set here it chases the same bytes to the same values.
    return !(err || k + 1 >= a.n_seg || a.entry[k + 1] != NONE);
}

// one lane per segment
__global__ __launch_bounds__(256) void k0_chase(FrameArgs a) {
    __shared__ uint16_t ring[RING_N][256];
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_seg) return;
    const uint32_t d = a.dirty[k];
    if (!d) return;
    if (d == 2) a.entry[k] = a.fix[k];                                     // repaired by k0_verify
    a.dirty[k] = 0;
    uint64_t off = a.entry[k];
    if (off == NONE) return;                                               // no guess: end/cnt/... keep their initial NONE/0 unless a chase from the left walks in (below)
    if (off == STOP) { a.end[k] = off; a.cnt[k] = 0; a.lseq[k] = 0; a.ncig[k] = 0; a.seg_err[k] = 0; return; }
    const RdGlobal rd{a.s};
    if (a.side16) { while (chase_one<RdGlobal, 1>(a, k, &off, rd, &ring[0][0])) { ++k; a.entry[k] = off; a.dirty[k] = 0; } }
    else { while (chase_one<RdGlobal, 0>(a, k, &off, rd)) { ++k; a.entry[k] = off; a.dirty[k] = 0; } }
}

// The repair rounds: k0_verify found a handful of segments whose guess was wrong (a byte pattern inside a record that passes for a
// header: a few in 10^5 segments) and listed them.  One lane chasing such a segment again through global memory is ~230 dependent
// misses, 0.18 ms -- 4 % of a step on 10 M reads -- whatever the number of segments; here a workgroup per listed segment copies
// it into LDS (coalesced) and one thread chases there.
__global__ __launch_bounds__(256) void k0_rechase(FrameArgs a, const uint32_t* list, uint32_t n_list) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rw[];
    __shared__ uint64_t s_off, s_end; __shared__ int s_go, s_plain; __shared__ uint32_t s_n;
    __shared__ unsigned long long s_acc[3 * 8 + 3];           // per window boundary: records, bases, ops in front of it; [24] a long CIGAR seen, [25] a record that fails its tests
    if (blockIdx.x >= n_list) return;
    uint64_t k = list[blockIdx.x];
    const int t = threadIdx.x;
    const uint32_t lst_cap = a.seg_bytes / 36u + 2u;
    uint32_t* const lst = reinterpret_cast<uint32_t*>(rw + (((size_t)a.seg_bytes + 96 + 15) & ~(size_t)15));      // the segment's record offsets (behind the window)
    if (t == 0) {
        a.entry[k] = a.fix[k]; a.dirty[k] = 0;
        s_off = a.entry[k]; s_go = 1;
        if (s_off == NONE) s_go = 0;
        else if (s_off == STOP) { a.end[k] = STOP; a.cnt[k] = 0; a.lseq[k] = 0; a.ncig[k] = 0; a.seg_err[k] = 0; s_go = 0; }
    }
    __syncthreads();
    while (s_go) {
        const uint64_t seg0 = k * (uint64_t)a.seg_bytes, seg_end = min(seg0 + a.seg_bytes, a.len);
        // from the entry to a header's length past the segment, whole 16-byte pieces -- and never more than the window holds: in a batch
        // (tolerant tail) the entry may lie in front of the segment, with a record still incomplete at the batch's end
        const uint64_t w0 = (s_off & ~15ull), w1 = min(min(a.len & ~15ull, ((k + 1) * (uint64_t)a.seg_bytes + 64) & ~15ull), w0 + (((uint64_t)a.seg_bytes + 96) & ~15ull));
        if (w1 > w0) for (uint64_t o = w0 + 16ull * t; o < w1; o += 16ull * 256) { uint4 v; __builtin_memcpy(&v, a.s + o, 16); *reinterpret_cast<uint4*>(rw + (o - w0)) = v; }
        if (t < 27) s_acc[t] = 0;
        __syncthreads();
        // One thread follows the length prefixes -- nothing but "read four bytes, add": the full hop of chase_one is ~100 dependent
        // instructions, 0.11 ms for a segment's 230 records by one thread -- and stops at anything that is not a plain hop; then
        // every record is looked at by a thread of its own.  Whatever is not plain (the end of the data, a record that fails a test)
        // is left to chase_one, in the reference's order of tests.
        if (t == 0) {
            uint32_t n = 0; uint64_t off = s_off; bool plain = a.sub_n <= 8;
            while (plain && off < seg_end) {
                if (off < seg0 || off < w0 || off + 24 > w1 || n >= lst_cap) { plain = false; break; }
                uint32_t bs; __builtin_memcpy(&bs, rw + (off - w0), 4);
                if ((int32_t)bs < 32 || (uint64_t)bs > a.len - off - 4) { plain = false; break; }
                lst[n++] = (uint32_t)(off - seg0);
                off += 4 + (uint64_t)bs;
            }
            s_n = n; s_plain = plain ? 1 : 0; s_end = off;
        }
        __syncthreads();
        if (s_plain) {
            const uint32_t n = s_n, nb = a.sub ? a.sub_n : 1u;
            for (uint32_t j = t; j < n; j += 256) {
                const uint32_t o = lst[j];
                const uint8_t* h = rw + (seg0 + o - w0);
                uint32_t bs, f3[3]; __builtin_memcpy(&bs, h, 4); __builtin_memcpy(f3, h + 12, 12);
                const int32_t lseq = (int32_t)f3[2];
                const uint32_t l_name = f3[0] & 0xFFu, n_cig = f3[1] & 0xFFFFu;
                if (lseq < 0 || l_name == 0 || 32ull + l_name + 4ull * n_cig + (uint64_t)((lseq + 1) / 2) + (uint64_t)lseq > (uint64_t)bs) { atomicAdd(&s_acc[25], 1ull); continue; }
                if (n_cig > LONG_CIGAR) atomicAdd(&s_acc[24], 1ull);
                if (n_cig > BGP_PLAN_MAX_OPS || lseq > 255) atomicAdd(&s_acc[26], 1ull);
                const uint32_t wj = a.sub ? o >> a.sub_shift : 0u;                     // the record counts in front of every window behind its own, and in the totals
                for (uint32_t w = wj + 1; w <= nb; ++w) { unsigned long long* q = s_acc + 3 * (w - 1); atomicAdd(q, 1ull); atomicAdd(q + 1, (unsigned long long)(uint32_t)lseq); atomicAdd(q + 2, (unsigned long long)n_cig); }
            }
            __syncthreads();
            if (s_acc[25]) { if (t == 0) s_plain = 0; }
            __syncthreads();
        }
        if (s_plain) {
            const uint32_t n = s_n, nb = a.sub ? a.sub_n : 1u;
            if (a.side16) { uint16_t* d16 = reinterpret_cast<uint16_t*>(a.side + k * a.side_cap); for (uint32_t j = t; j < n && j < 2u * a.side_cap; j += 256) d16[j] = (uint16_t)lst[j]; }
            else { uint32_t* d32 = a.side + k * a.side_cap; for (uint32_t j = t; j < n && j < a.side_cap; j += 256) d32[j] = lst[j]; }
            if (a.sub && t >= 1 && (uint32_t)t < nb) a.sub[k * a.sub_n + t] = make_uint4((uint32_t)s_acc[3 * (t - 1)], (uint32_t)s_acc[3 * (t - 1) + 1], (uint32_t)s_acc[3 * (t - 1) + 2], 0u);
            if (t == 0) {
                const unsigned long long* tot = s_acc + 3 * (nb - 1);
                a.end[k] = s_end; a.cnt[k] = tot[0]; a.lseq[k] = tot[1]; a.ncig[k] = tot[2]; a.seg_err[k] = (s_acc[24] ? ST_LONGCIG : 0u) | (s_acc[26] ? ST_ODDREAD : 0u);
                const bool on = !(k + 1 >= a.n_seg || a.entry[k + 1] != NONE);
                if (on) { a.entry[k + 1] = s_end; a.dirty[k + 1] = 0; }
                s_off = s_end; s_go = on ? 1 : 0;
            }
        } else if (t == 0) {
            uint64_t off = s_off;
            const RdLds rd{a.s, rw, w0, w1 > w0 ? w1 : w0};
            const bool on = a.side16 ? chase_one<RdLds, 2>(a, k, &off, rd) : chase_one<RdLds, 0>(a, k, &off, rd);
            if (on) { a.entry[k + 1] = off; a.dirty[k + 1] = 0; }
            s_off = off; s_go = on ? 1 : 0;
        }
        __syncthreads();
        ++k;
    }
}

// A link k is consistent when entry[k] == end[k-1].  A broken link is repaired only when the
// predecessor's own link is consistent: otherwise end[k-1] was chased from a wrong entry and would
// overwrite a (probably correct) guess with garbage, which then ripples forward one segment per
// round forever.  entry[] is never written here (fix[] is applied by the next k0_chase), so every
// thread judges its neighbours from the same snapshot.
__global__ __launch_bounds__(256) void k0_verify(FrameArgs a) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (k >= a.n_seg) return;
    const uint64_t prev_end = a.end[k - 1];
    if (a.entry[k] == prev_end) return;
    if (k >= 2 && a.entry[k - 1] != a.end[k - 2]) return;
    a.fix[k] = prev_end;
    a.dirty[k] = 2;
    const uint32_t slot = atomicAdd(&a.status[1], 1u);
    if (slot < a.list_cap) a.list[slot] = (uint32_t)k;
}

__global__ __launch_bounds__(256) void k0_collect_errors(FrameArgs a) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_seg) return;
    const uint32_t e = a.seg_err[k];
    if (e) atomicOr(&a.status[0], e);
}

struct EmitArgs {
    const uint8_t* s; uint64_t len; uint32_t seg_bytes; uint64_t n_seg;
    const uint64_t* entry; const uint64_t* rec_base; const uint64_t* seq_base; const uint64_t* cig_base;
    bgp_record* rec; uint64_t* seq_off; uint64_t* cigar_off; uint4* cig4; uint64_t n_records, sum_lseq, sum_ncig;
};

#ifdef BGP_EXPERIMENTS
// one lane per segment: second (L2-warm) chase that writes the record table
__global__ __launch_bounds__(256) void k1_emit(EmitArgs a) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) { a.seq_off[a.n_records] = a.sum_lseq; a.cigar_off[a.n_records] = a.sum_ncig; }
    if (k >= a.n_seg) return;
    uint64_t off = a.entry[k];
    if (off == NONE || off == STOP) return;
    const uint64_t seg_end = min((k + 1) * (uint64_t)a.seg_bytes, a.len);
    uint64_t r = a.rec_base[k], so = a.seq_base[k], co = a.cig_base[k];
    const uint64_t r_end = a.rec_base[k + 1];
    while (off < seg_end && r < r_end) {
        const uint8_t* p = a.s + off;
        const uint32_t bs = ld_u32(p);
        bgp_record rec;
        rec.ref_id = (int32_t)ld_u32(p + 4);                     // read.d:907
        rec.pos = (int32_t)ld_u32(p + 8);                        // read.d:911
        rec.bin_mq_nl = ld_u32(p + 12);                          // read.d:915
        rec.flag_nc = ld_u32(p + 16);                            // read.d:919
        rec.l_seq = (int32_t)ld_u32(p + 20);                     // read.d:923
        rec.rec_off = off;
        const uint32_t l_name = rec.bin_mq_nl & 0xFFu, n_cig = rec.flag_nc & 0xFFFFu;
        uint32_t covered = 0;
        uint32_t c4[4] = {0, 0, 0, 0};
        if (!((rec.flag_nc >> 16) & 0x4u) && n_cig <= LONG_CIGAR) { // read.d:257-259 is_unmapped -> 0; long CIGARs: k1_long_cigar
            const uint8_t* cg = p + 36 + l_name;                  // read.d:988-990 _cigar_offset
            for (uint32_t j = 0; j < n_cig; ++j) {
                const uint32_t op = ld_u32(cg + 4 * j);
                covered += cig_rc(op) ? (op >> 4) : 0u;           // read.d:261
                if (j < 4) c4[j] = op;
            }
        }
        a.cig4[r] = make_uint4(c4[0], c4[1], c4[2], c4[3]);
        rec.end_pos = (int32_t)((uint32_t)rec.pos + covered);     // read.d:1382
        a.rec[r] = rec;
        a.seq_off[r] = so;
        a.cigar_off[r] = co;
        so += (uint64_t)rec.l_seq; co += n_cig; ++r;
        off += 4 + (uint64_t)bs;
    }
}
#endif

// basesCovered() of records with long CIGARs (read.d:255-262): a lane summing thousands of ops serially
// stalls its whole wave in k1_emit, so those records are finished here, one wave per record, 64 ops per
// coalesced load and a wave reduction.
// rpw: records a wave looks after (its first lanes hold them).  A wave walks its long records one after the other, so a stream of
// few, long reads wants few records a wave (cfg5: 200 k records are 3125 waves of 64 -- a chain of 64 x 8 round trips each on a device
// with 8192 wave slots -- or 25 000 waves of 8); a stream of 10^8 short reads with the odd long CIGAR wants 64.
__global__ __launch_bounds__(256) void k1_long_cigar(const uint8_t* s, bgp_record* rec, uint64_t n, uint32_t rpw) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t i = wave * rpw + (uint32_t)lane;
    bgp_record r{};
    bool lng = false;
    if ((uint32_t)lane < rpw && i < n) { r = rec[i]; lng = (r.flag_nc & 0xFFFFu) > LONG_CIGAR && !((r.flag_nc >> 16) & 0x4u); }
    unsigned long long m = __ballot(lng);
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
        const uint32_t n_cig = (uint32_t)__shfl((int)r.flag_nc, src, 64) & 0xFFFFu, l_name = (uint32_t)__shfl((int)r.bin_mq_nl, src, 64) & 0xFFu;
        const uint64_t rec_off = ((uint64_t)(uint32_t)__shfl((int)(r.rec_off >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)r.rec_off, src, 64);
        const uint8_t* cg = s + rec_off + 36 + l_name;
        uint32_t acc = 0;
        // eight loads a lane in flight: one at a time, a 4000-op CIGAR was 62 dependent round trips a record and a wave walks its 64
        // records one after the other (cfg5: 1.1 ms for 3.2 GB of ops)
        uint32_t j = lane;
        for (; j + 64u * 7u < n_cig; j += 64u * 8u) {
            uint32_t op[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) op[u] = ld_u32(cg + 4ull * (j + 64u * u));
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cig_rc(op[u]) ? (op[u] >> 4) : 0u;
        }
        for (; j < n_cig; j += 64) { const uint32_t op = ld_u32(cg + 4ull * j); acc += cig_rc(op) ? (op >> 4) : 0u; }
        for (int d = 32; d >= 1; d >>= 1) acc += (uint32_t)__shfl_xor((int)acc, d, 64);
        if (lane == src) r.end_pos = (int32_t)((uint32_t)r.pos + acc);     // read.d:1382
    }
    if (lng) rec[i].end_pos = r.end_pos;
}

uint32_t pow2_ceil(uint64_t v) { uint32_t p = 1; while (p < v && p < (1u << 30)) p <<= 1; return p; }

}  // namespace

int k0_frame(bgp_ctx* c) {
    const uint64_t len = c->stream_len;
    c->n_records = 0; c->sum_lseq = 0; c->sum_ncig = 0;
    BGP_TRY(c->d_status.reserve(64));
    BGP_HIP(hipMemsetAsync(c->d_status.p, 0, 64, c->stream));
    if (len < 4) {                       // empty range (readrange.d:123-149)
        BGP_TRY(c->d_rec.reserve(sizeof(bgp_record)));
        BGP_TRY(c->d_seq_off.reserve(8)); BGP_TRY(c->d_cigar_off.reserve(8));
        BGP_HIP(hipMemsetAsync(c->d_seq_off.p, 0, 8, c->stream));
        BGP_HIP(hipMemsetAsync(c->d_cigar_off.p, 0, 8, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        c->table_pending = false; c->long_cigar = false;
        c->frame_tail = c->frame_tolerant_tail ? 0 : len;      // (a batch keeps the few bytes for the next one)
        return BGP_OK;
    }
    // segment size: ~8 average records, from the first record's size (any value is correct)
    // (a host round trip of its own -- 30 us on a quiet host, 200 on a busy one -- unless these very bytes were framed before: a consumer that
    // frames a resident stream again finds the value the last pass read; every pass reads it afresh with its last fetch, below)
    uint32_t bs0 = 0;
    if (c->bs0_hint && c->bs0_ptr == c->d_stream && c->bs0_len == len) bs0 = c->bs0_hint;
    else { const MailItem it{c->d_stream, &bs0, 4}; BGP_TRY(mail_fetch(c, &it, 1)); }
    // ~64 records per segment: k0_guess costs one wave per segment, the chase is a per-lane serial chain per
    // segment that is bound by line fetches, not by its length (measured on cfg3, guess / chase / step: 8 records
    // 7.4 / 1.2 ms; 32: 1.87 / 1.58 / 43.8 ms; 64: 0.94 / 1.60 / 42.5 ms; 128: 0.48 / 1.66 / 43.0 ms)
    const uint64_t recs_per_seg = getenv("BGP_SEG_RECORDS") ? (uint64_t)atoi(getenv("BGP_SEG_RECORDS")) : 64ull;
    uint64_t want = recs_per_seg * (4ull + (bs0 & 0x3FFFFFFFu));
    uint32_t seg = pow2_ceil(want);
    if (seg < 4096u) seg = 4096u;
    if (seg > (1u << 22)) seg = 1u << 22;            // (long reads: every segment costs a guess that walks half a record; 1 MiB segments made that 2.2 ms of cfg5, 4 MiB 0.6)
    if (c->seg_bytes_wanted && seg <= c->seg_bytes_wanted) seg = c->seg_bytes_wanted;     // the LDS-window decode wants its own segment size (any value is correct)
    c->seg_bytes = seg;
    const uint64_t n_seg = (len + seg - 1) / seg;
    c->n_seg = n_seg;

    BGP_TRY(c->d_seg_entry.reserve((n_seg + 1) * 16)); BGP_TRY(c->d_seg_end.reserve((n_seg + 1) * 8));
    BGP_TRY(c->d_seg_cnt.reserve((n_seg + 1) * 8)); BGP_TRY(c->d_seg_lseq.reserve((n_seg + 1) * 8)); BGP_TRY(c->d_seg_ncig.reserve((n_seg + 1) * 8));
    BGP_TRY(c->d_seg_dirty.reserve((n_seg + 1) * 8));          // dirty[n_seg] u32 followed by seg_err[n_seg] u32
    BGP_TRY(c->d_seg_rec_base.reserve((n_seg + 2) * 8)); BGP_TRY(c->d_seg_seq_base.reserve((n_seg + 2) * 8)); BGP_TRY(c->d_seg_cig_base.reserve((n_seg + 2) * 8));

    FrameArgs a;
    a.s = c->d_stream; a.len = len; a.seg_bytes = seg; a.n_seg = n_seg;
    a.ref_bound = c->n_ref_hint > 0 ? c->n_ref_hint : (1 << 24);
    a.tolerant_tail = c->frame_tolerant_tail ? 1 : 0;
    a.sub = nullptr; a.sub_n = 0; a.sub_shift = 0; c->seg_windows = 0;
    a.seg_hint = nullptr;
    constexpr uint32_t LIST_CAP = 1024;
    BGP_TRY(c->d_seg_list.reserve(LIST_CAP * 4));
    a.list = c->d_seg_list.as<uint32_t>(); a.list_cap = n_seg < (1ull << 32) ? LIST_CAP : 0u;
    a.entry = c->d_seg_entry.as<uint64_t>(); a.fix = a.entry + n_seg + 1; a.end = c->d_seg_end.as<uint64_t>();
    a.cnt = c->d_seg_cnt.as<uint64_t>(); a.lseq = c->d_seg_lseq.as<uint64_t>(); a.ncig = c->d_seg_ncig.as<uint64_t>();
    a.dirty = c->d_seg_dirty.as<uint32_t>(); a.seg_err = a.dirty + n_seg;
    a.status = c->d_status.as<uint32_t>();
    a.side_cap = (seg / 36u + 2u + 2u * SIDE_GROUP - 1u) / SIDE_GROUP * SIDE_GROUP;      // a record is at least 36 bytes (block_size >= 32 is enforced by the chase); whole groups, the last one may be written half empty
    if (c->seg_bytes_wanted && seg == c->seg_bytes_wanted && c->seg_window_shift) {      // k12_lds follows: its windows' prefix counts inside every segment
        a.sub_shift = c->seg_window_shift; a.sub_n = seg >> a.sub_shift;
        BGP_TRY(c->d_seg_sub.reserve((n_seg + 1) * (size_t)a.sub_n * 16));
        BGP_HIP(hipMemsetAsync(c->d_seg_sub.p, 0, (n_seg + 1) * (size_t)a.sub_n * 16, c->stream));
        a.sub = c->d_seg_sub.as<uint4>(); c->seg_windows = a.sub_n;
    }
    BGP_TRY(c->d_seg_side.reserve((n_seg + 1) * (size_t)a.side_cap * 4));
    a.side = c->d_seg_side.as<uint32_t>();
    c->side_cap = a.side_cap;
    a.side16 = a.sub != nullptr && seg <= 65536u; c->side16 = a.side16 != 0;

    if (!c->hints.empty()) {
        std::vector<uint64_t> sh(n_seg, NONE);
        for (uint64_t h : c->hints) if (h < len) { uint64_t& e = sh[h / seg]; if (h < e) e = h; }
        BGP_TRY(c->d_seg_hint.reserve(n_seg * 8));
        BGP_HIP(hipMemcpyAsync(c->d_seg_hint.p, sh.data(), n_seg * 8, hipMemcpyHostToDevice, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
        a.seg_hint = c->d_seg_hint.as<uint64_t>();
    }
    const unsigned lane_blocks = (unsigned)((n_seg + 255) / 256);
    {
        TimedRegion t(c, "k0_guess", n_seg * 64);
        {   // every segment dirty (any non-zero); what a segment without an entry keeps: end NONE, no records
            const FillItem f[6] = {{a.dirty, n_seg * 4, 0x01u}, {a.seg_err, n_seg * 4, 0u}, {a.end, n_seg * 8, 0xFFu}, {a.cnt, n_seg * 8, 0u}, {a.lseq, n_seg * 8, 0u}, {a.ncig, n_seg * 8, 0u}};
            if (((uintptr_t)a.seg_err & 15) == 0) BGP_TRY(fill_many(c, f, 6));
            else {      // (seg_err sits behind dirty[n_seg] in one buffer: a multiple of four segments keeps it aligned)
                BGP_HIP(hipMemsetAsync(a.dirty, 0x01, n_seg * 4, c->stream)); BGP_HIP(hipMemsetAsync(a.seg_err, 0, n_seg * 4, c->stream));
                BGP_TRY(fill_many(c, f + 2, 4));
            }
        }
        if (n_seg <= 32768) hipLaunchKernelGGL(k0_guess_wide, dim3((unsigned)n_seg), dim3(256), 0, c->stream, a);      // few, large segments: a workgroup each
        else hipLaunchKernelGGL(k0_guess, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0, c->stream, a);
    }
    uint32_t bits = 0;
    uint64_t n = 0, sl = 0, sc = 0, tail = len;      // (what the last round's fetch brings: status bits, the totals of the three scans, where the chain stopped)
    uint64_t rounds = 0;
    uint32_t n_list = 0;                 // the segments of a repair round, when they are few enough to be listed and small enough for LDS
    // k0_rechase keeps a segment and a ninth more in LDS -- beyond the 64 KiB a kernel gets without asking: whether the device grants that is
    // asked ONCE per framing (a refusal used to be swallowed round after round, with k0_chase's time booked under k0_rechase's name)
    const size_t rechase_lds = (((size_t)seg + 96 + 15) & ~(size_t)15) + 4 * ((size_t)seg / 36 + 2) + 16;
    bool rechase_ok = seg <= (128u << 10);
    if (rechase_ok && rechase_lds > (48u << 10)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k0_rechase), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rechase_lds) != hipSuccess) { (void)hipGetLastError(); rechase_ok = false; }
    }
    for (;;) {
        {
            const bool listed = n_list != 0 && rechase_ok;
            TimedRegion t(c, listed ? "k0_rechase" : "k0_chase", 0);
            if (listed) { hipLaunchKernelGGL(k0_rechase, dim3(n_list), dim3(256), rechase_lds, c->stream, a, a.list, n_list); BGP_HIP(hipGetLastError()); }
            else { hipLaunchKernelGGL(k0_chase, dim3(lane_blocks), dim3(256), 0, c->stream, a); BGP_HIP(hipGetLastError()); }
        }
        BGP_HIP(hipMemsetAsync(a.status, 0, 8, c->stream));      // the status bits and the dirty counter are this round's
        hipLaunchKernelGGL(k0_verify, dim3(lane_blocks), dim3(256), 0, c->stream, a);
        // What follows the LAST round -- the segments' error bits, the three exclusive scans (records, bases, CIGAR ops in front of every
        // segment), where the chain stopped -- is queued behind EVERY round and comes back with the round's dirty count in one round trip:
        // a round that leaves nothing dirty (the second, as a rule) is the last, and its results are already here.  A round that does has
        // run three small scans for nothing (~20 us of kernels against a host round trip of 30 us on a quiet host and 200 on a busy one).
        hipLaunchKernelGGL(k0_collect_errors, dim3(lane_blocks), dim3(256), 0, c->stream, a);
        {
            const uint64_t* const in[3] = {a.cnt, a.lseq, a.ncig};
            uint64_t* const out[3] = {c->d_seg_rec_base.as<uint64_t>(), c->d_seg_seq_base.as<uint64_t>(), c->d_seg_cig_base.as<uint64_t>()};
            BGP_TRY(scan_exclusive_u64x3(c, in, out, n_seg, nullptr));
        }
        uint32_t ndirty = 0, first_bs = 0;
        {   // (tail: where the chain of records stopped -- in a batch the bytes behind it belong to the next one; first_bs: the first record's block_size, for the next pass over these bytes)
            const MailItem it[7] = {{a.status + 1, &ndirty, 4}, {a.status, &bits, 4}, {c->d_seg_rec_base.as<uint64_t>() + n_seg, &n, 8}, {c->d_seg_seq_base.as<uint64_t>() + n_seg, &sl, 8},
                                    {c->d_seg_cig_base.as<uint64_t>() + n_seg, &sc, 8}, {a.end + (n_seg - 1), &tail, 8}, {c->d_stream, &first_bs, 4}};
            BGP_TRY(mail_fetch(c, it, 7));
        }
        c->bs0_hint = first_bs; c->bs0_ptr = c->d_stream; c->bs0_len = len;
        if (getenv("BGP_VERBOSE") && rounds < 8) fprintf(stderr, "[biodgpu] framing round %llu: %u of %llu segments re-entered (seg %u B)\n", (unsigned long long)rounds, ndirty, (unsigned long long)n_seg, seg);
        if (ndirty == 0) break;
        n_list = ndirty <= a.list_cap && seg <= (128u << 10) ? ndirty : 0u;
        if (++rounds > n_seg + 2) { c->last_error = "record framing did not converge"; return BGP_EDEVICE; }
    }
    c->frame_tail = len;
    if (c->frame_tolerant_tail) c->frame_tail = tail;
    const int st = bgp_status_from_bits(bits);
    if (st < 0 && st != BGP_ESEQ) return st;
    c->n_records = n; c->sum_lseq = sl; c->sum_ncig = sc;
    if (c->frame_tail > len) c->frame_tail = len;
    if (c->timing)                       // what framing has to read: block_size and the 32 fixed bytes of every record (known only now)
        for (auto& t : c->timers) if (t.name == "k0_chase") t.algorithmic_bytes += n * 36;

    c->long_cigar = (bits & ST_LONGCIG) != 0; c->odd_reads = (bits & ST_ODDREAD) != 0;
    c->table_pending = true;             // the record table is written by whoever needs it first: k1_table, or the fused K1+K2 when a decode follows
    return BGP_OK;
}

#ifdef BGP_EXPERIMENTS
// K1 alone: the record table and arena offsets (second, L2-warm chase).  Used when the table is needed
// before -- or without -- a decode; bgp_decode_* right after bgp_flush takes the fused kernel in decode.hip.
int k1_table(bgp_ctx* c) {
    const uint64_t n = c->n_records, n_seg = c->n_seg, sl = c->sum_lseq, sc = c->sum_ncig;
    c->table_pending = false;
    if (c->stream_len < 4) return BGP_OK;
    const unsigned lane_blocks = (unsigned)((n_seg + 255) / 256);
    BGP_TRY(c->d_rec.reserve((n + 1) * sizeof(bgp_record)));
    BGP_TRY(c->d_seq_off.reserve((n + 1) * 8)); BGP_TRY(c->d_cigar_off.reserve((n + 1) * 8));
    BGP_TRY(c->d_cig4.reserve((n + 1) * 16)); c->cig4_ok = true; c->rows_ok = false; c->lds_table = false; c->tilemax_ok = false;
    EmitArgs e;
    e.s = c->d_stream; e.len = c->stream_len; e.seg_bytes = c->seg_bytes; e.n_seg = n_seg; e.entry = c->d_seg_entry.as<uint64_t>();
    e.rec_base = c->d_seg_rec_base.as<uint64_t>(); e.seq_base = c->d_seg_seq_base.as<uint64_t>(); e.cig_base = c->d_seg_cig_base.as<uint64_t>();
    e.rec = c->d_rec.as<bgp_record>(); e.seq_off = c->d_seq_off.as<uint64_t>(); e.cigar_off = c->d_cigar_off.as<uint64_t>(); e.cig4 = c->d_cig4.as<uint4>();
    e.n_records = n; e.sum_lseq = sl; e.sum_ncig = sc;
    {
        TimedRegion t(c, "k1_emit", n * (36 + 32 + 16) + sc * 4);
        hipLaunchKernelGGL(k1_emit, dim3(lane_blocks), dim3(256), 0, c->stream, e);
        if (c->long_cigar)                // only streams that hold a long CIGAR pay for the extra pass
            hipLaunchKernelGGL(k1_long_cigar, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_stream, e.rec, n, 64u);
    }
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}
#endif

// basesCovered() of long-CIGAR records after the fused K1+K2 wrote the table (same pass k1_table runs)
int k1_finish_long_cigars(bgp_ctx* c) {
    if (!c->long_cigar || !c->n_records) return BGP_OK;
    if (c->defer_long_cigar) { c->long_cigar_deferred = true; return BGP_OK; }      // (k2_decode does it on its way through the CIGARs)
    const uint32_t rpw = c->n_records < (4ull << 20) ? 8u : 64u;
    const uint64_t waves = (c->n_records + rpw - 1) / rpw;
    hipLaunchKernelGGL(k1_long_cigar, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, c->stream, c->d_stream, c->d_rec.as<bgp_record>(), c->n_records, rpw);
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}
