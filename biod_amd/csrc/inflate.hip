// inflate.hip -- BGZF blocks inflated on the device ("device inflate").
//
// Reference semantics: decompressBgzfBlock (bio/core/bgzf/block.d:127-216) -- raw DEFLATE (RFC 1951) of the block's
// CDATA into input_size bytes, then the size check and the CRC32 check against the block footer; BioD runs it on a
// TaskPool, one task per block (bgzf/inputstream.d:386-447).  Here one WAVE takes one block: blocks are independent
// (the LZ77 window never crosses a BGZF block), a 100 M-read file has ~230 k of them, and the host side of a GPU
// box is small (the measured box: 256 logical CPUs under a 16-CPU quota, 5 GB/s of zlib at best).
//
// Inside a wave the Huffman decode is serial by nature; everything around it is spread over the 64 lanes:
//   * compressed bytes: every lane holds one dword of a 256-byte chunk; the bit buffer refills with v_readlane
//     (no memory latency on the decode chain), the next chunk is one coalesced load;
//   * canonical codes without lookup tables: lane L (1..15) holds first code / count / symbol offset of length L;
//     one symbol = bit-reverse 15 peeked bits, every lane tests its own length, ballot + ffs picks the shortest
//     hit, one LDS read fetches the symbol.  Building a code is 15 ballots per 64 symbols (counts, then ranks);
//   * literals collect in a lane register and leave as one 64-byte store; a match is copied by all lanes at once
//     (source index modulo the distance, so overlapping runs need no loop-carried dependency);
//   * CRC32: every lane takes a 1/64 slice, the slice CRCs are shifted by x^(8*bytes after) in GF(2)[x]/P and
//     xor-reduced (zlib's crc32_combine arithmetic).
//   * short codes (<= 10 bits literal/length, <= 9 bits distance) resolve through a per-wave LDS table filled by
//     decoding every prefix once in parallel; longer codes take the lane-parallel path above.
// All decoder state is wave-uniform and lives in SGPRs (the wave index is passed through readfirstlane so the
// compiler can see that).  What bounds the kernel is the serial chain per symbol (table lookup -> readfirstlane ->
// scalar shifts -> branch), not memory: measured 42 GB/s of inflated bytes on a generator file (45 k blocks, 1.4 G
// symbols, 69 ms), 8x the 16-core host it replaces; DESIGN.md lists what did and did not move it.  Two waits mattered:
// on gfx9 one vmcnt counter covers loads and stores, so the input-chunk load is consumed inside its own branch
// (not at every refill), and a match loads its source before the pending literals are stored.
// Malformed input ends in a status code, never in a write outside [dst, dst + isize): every store is bounds-checked
// against the footer's ISIZE, every loop consumes input bits or produces output.
#include "common.h"

namespace {

constexpr uint32_t POLY = 0xEDB88320u;

struct CrcTables { uint32_t t[256]; uint32_t x2n[32]; };

constexpr uint32_t multmodp_c(uint32_t a, uint32_t b) {            // a(x) * b(x) mod P, reflected (zlib crc32.c multmodp)
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ POLY : b >> 1;
    }
    return p;
}
constexpr CrcTables make_crc_tables() {
    CrcTables r{};
    for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ POLY : c >> 1; r.t[i] = c; }
    uint32_t p = 1u << 30;                                         // x^1
    r.x2n[0] = p;
    for (int n = 1; n < 32; ++n) { p = multmodp_c(p, p); r.x2n[n] = p; }
    return r;
}
__constant__ CrcTables CRC = make_crc_tables();

__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

constexpr int WAVES = 4;
struct WaveLds { uint16_t llut[1024]; uint16_t dlut[512]; uint16_t lsym[288]; uint16_t dsym[32]; uint16_t csym[20]; uint8_t lens[352]; uint32_t lcode[48], dcode[48]; };   // lcode/dcode: the two codes' first/count/offs per length, for the long-code path

enum : int32_t { INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_BAD_STORED = 2, INF_BAD_CODE = 3, INF_BAD_LENGTHS = 4, INF_OVERRUN_OUT = 5, INF_BAD_DISTANCE = 6,
                 INF_OVERRUN_IN = 7, INF_SHORT_OUT = 8, INF_BAD_CRC = 9, INF_OVERSUBSCRIBED = 10 };

#ifdef BGP_INFLATE_STATS
__device__ unsigned long long g_stats[8];   // matches, dist <= 1k / 2k / 4k / 8k / 16k / 32k, match bytes
#endif

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rdl(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// The decoder's state is wave-uniform, and left alone the compiler keeps all of it in SGPRs: the profile then shows
// the scalar pipe saturated (one SALU issue per SIMD every four cycles, 87 % busy) next to an idle vector pipe.  So
// the bit buffer is made to LOOK lane-varying (every refill ORs in a zero the compiler cannot see through): shifts,
// masks and table addresses become VALU work that issues beside the scalar control flow.
__device__ __forceinline__ uint32_t opaque_zero() { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }

struct Bits {                                       // wave-uniform bit reader over aligned dwords
    const uint32_t* base; uint32_t n_dw; int lane; uint32_t vz;
    uint32_t chunk; int cur;                        // per lane: dword (cur * 64 + lane)
    uint64_t bb; int nb; uint32_t widx;
    __device__ __forceinline__ uint64_t bitpos() const { return (uint64_t)widx * 32u - rfl((uint32_t)nb); }   // bits consumed, from the aligned base
    __device__ __forceinline__ uint32_t fetch() {
        const int ch = (int)(widx >> 6);
        if (ch != cur) {
            const uint32_t i = (uint32_t)ch * 64u + (uint32_t)lane;
            const uint32_t got = i < n_dw ? base[i] : 0u;
            // the copy makes the wait for this load happen HERE, once per 256 input bytes; without it the wait sits at
            // every refill after the merge, and on gfx9 that vmcnt(0) also waits for every output store in flight
            asm volatile("v_mov_b32 %0, %1" : "=v"(chunk) : "v"(got));
            cur = ch;
        }
        const uint32_t w = rdl(chunk, widx & 63u);
        ++widx;
        return w;
    }
    __device__ __forceinline__ void need32() { if ((int)rfl((uint32_t)nb) <= 32) { bb |= (uint64_t)(fetch() | vz) << nb; nb += 32; } }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = rfl((uint32_t)bb & ((1u << n) - 1u)); bb >>= n; nb -= n; return v; }
    __device__ __forceinline__ void seek(uint64_t bit) {
        widx = (uint32_t)(bit >> 5); nb = (int)vz; bb = vz;
        need32();
        (void)take((int)(bit & 31u));
    }
};

struct Code { uint32_t first, count, offs; };       // lane L (1..15): codes of length L
// the literal/length and distance codes are only needed per lane when a code is longer than the table covers:
// they wait in LDS instead of holding six VGPRs through the hot loop
__device__ __forceinline__ void park_code(const Code& cd, uint32_t* at, int lane) { if (lane < 16) { at[lane] = cd.first; at[16 + lane] = cd.count; at[32 + lane] = cd.offs; } }
__device__ __forceinline__ Code fetch_code(const uint32_t* at, int lane) { Code cd; cd.first = at[lane & 15]; cd.count = lane < 16 ? at[16 + lane] : 0u; cd.offs = at[32 + (lane & 15)]; return cd; }

// lens[0..n) -> canonical code in lane registers + symbols sorted by (length, value) in sym.  false: over-subscribed.
__device__ bool build_code(const uint8_t* lens, int n, uint16_t* sym, Code& cd, int lane) {
    uint32_t cnt = 0;
    for (int b = 0; b < n; b += 64) {
        const int s = b + lane;
        const int L = s < n ? lens[s] : 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) { const uint64_t m = __ballot(L == l); if (lane == l) cnt += (uint32_t)__popcll(m); }
    }
    uint32_t o = 0, code = 0, myo = 0, myf = 0; int left = 1; bool ok = true;
#pragma unroll
    for (int l = 1; l <= 15; ++l) {
        const uint32_t c = rdl(cnt, (uint32_t)l);
        if (lane == l) { myo = o; myf = code; }
        o += c; code = (code + c) << 1;
        left = (left << 1) - (int)c; ok = ok && left >= 0;
    }
    cd.first = myf; cd.count = (lane >= 1 && lane <= 15) ? cnt : 0u; cd.offs = myo;
    uint32_t run = myo;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int b = 0; b < n; b += 64) {
        const int s = b + lane;
        const int L = s < n ? lens[s] : 0;
#pragma unroll
        for (int l = 1; l <= 15; ++l) {
            const uint64_t m = __ballot(L == l);
            const uint32_t at = rdl(run, (uint32_t)l);
            if (L == l) sym[at + (uint32_t)__popcll(m & lt)] = (uint16_t)s;
            if (lane == l) run += (uint32_t)__popcll(m);
        }
    }
    return ok;
}

// one symbol: needs >= 15 bits in the buffer (zero-filled past the end of input).  ~0u: no code matches.
__device__ __forceinline__ uint32_t decode_sym(Bits& B, const Code& cd, const uint16_t* sym, int lane) {
    const uint32_t rev = __brev(rfl((uint32_t)B.bb) & 0x7FFFu) >> 17;     // first stream bit = most significant of 15
    const uint32_t c = rev >> (15 - (lane & 15));
    const bool hit = lane >= 1 && lane <= 15 && (c - cd.first) < cd.count;
    const uint64_t m = __ballot(hit);
    if (!m) return ~0u;
    const uint32_t len = (uint32_t)__ffsll((unsigned long long)m) - 1u;
    const uint32_t idx = rdl(cd.offs + c - cd.first, len);
    const uint32_t s = rfl(sym[idx]);
    (void)B.take((int)len);
    return s;
}

// Every BITS-bit prefix decoded once, in parallel: lut[p] = symbol << 4 | code length for the code that p starts with
// (p in stream order, first bit lowest), 0 where the code is longer than BITS -- those go through decode_sym.
template <int BITS> __device__ void fill_lut(const Code& cd, const uint16_t* sym, uint16_t* lut, int lane) {
    uint32_t first[BITS + 1], count[BITS + 1], offs[BITS + 1];
#pragma unroll
    for (int l = 1; l <= BITS; ++l) { first[l] = rdl(cd.first, (uint32_t)l); count[l] = rdl(cd.count, (uint32_t)l); offs[l] = rdl(cd.offs, (uint32_t)l); }
    for (int p = lane; p < (1 << BITS); p += 64) {
        const uint32_t rev = __brev((uint32_t)p) >> (32 - BITS);
        uint32_t len = 0, idx = 0;
#pragma unroll
        for (int l = BITS; l >= 1; --l) {                          // the shortest hit wins: walk down, overwrite
            const uint32_t c = (rev >> (BITS - l)) - first[l];
            if (c < count[l]) { len = (uint32_t)l; idx = offs[l] + c; }
        }
        lut[p] = len ? (uint16_t)((sym[idx] << 4) | len) : (uint16_t)0;
    }
}

constexpr int LBITS = 10, DBITS = 9;

#ifndef BGP_INF_WAVES_PER_SIMD
#define BGP_INF_WAVES_PER_SIMD 8   // 64 VGPRs (a few spills): once the scalar pipe stopped being the limit, occupancy became it (5: 90 ms, 6: 80, 7: 74, 8: 69)
#endif
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(BGP_INF_WAVES_PER_SIMD, BGP_INF_WAVES_PER_SIMD))) void k_inflate(const uint8_t* __restrict__ comp, uint8_t* out, const bgp_inflate_block* __restrict__ blk, uint32_t n_blk, int32_t* status) {
    __shared__ WaveLds lds_all[WAVES];
    __shared__ uint32_t crc_t[256];
    crc_t[threadIdx.x] = CRC.t[threadIdx.x];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // the compiler cannot know tid >> 6 is wave-uniform
    const uint32_t bi = blockIdx.x * WAVES + wv;
    if (bi >= n_blk) return;
    WaveLds& lds = lds_all[wv];
    const bgp_inflate_block bk = blk[bi];
    const uint32_t isize = bk.isize;
    uint8_t* outp = out + bk.dst;
    int32_t err = INF_OK;

    Bits B;
    const uint64_t a0 = bk.src & ~3ull;
    B.base = (const uint32_t*)(comp + a0); B.n_dw = (uint32_t)(((bk.src & 3u) + bk.clen + 3u) >> 2); B.lane = lane; B.chunk = 0; B.cur = -1; B.vz = opaque_zero();
    B.seek((bk.src & 3u) * 8u);
    const uint64_t end_bit = ((bk.src & 3u) + (uint64_t)bk.clen) * 8u;

    uint32_t pos = 0, npend = 0, lit = 0;
    auto flush = [&]() {
        if (npend) {
            if (pos + npend > isize) { err = INF_OVERRUN_OUT; npend = 0; return; }
            if ((uint32_t)lane < npend) outp[pos + lane] = (uint8_t)lit;
            pos += npend; npend = 0;
        }
    };

    for (bool last = false; !last && !err;) {
        if (B.bitpos() > end_bit) { err = INF_OVERRUN_IN; break; }
        B.need32();
        last = B.take(1);
        const uint32_t type = B.take(2);
        if (type == 0) {                                           // stored
            flush();
            (void)B.take((int)((8u - (uint32_t)(B.bitpos() & 7u)) & 7u));
            B.need32();
            const uint32_t len = B.take(16);
            B.need32();
            const uint32_t nlen = B.take(16);
            if ((len ^ nlen) != 0xFFFFu) { err = INF_BAD_STORED; break; }
            const uint64_t from = a0 + (B.bitpos() >> 3);
            if (B.bitpos() + (uint64_t)len * 8u > end_bit) { err = INF_OVERRUN_IN; break; }
            if (pos + len > isize) { err = INF_OVERRUN_OUT; break; }
            for (uint32_t k = lane; k < len; k += 64) outp[pos + k] = comp[from + k];
            pos += len;
            B.seek(B.bitpos() + (uint64_t)len * 8u);
            continue;
        }
        if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
        Code lc{}, dc{};
        if (type == 1) {                                           // fixed code, RFC 1951 3.2.6
            for (int s = lane; s < 288; s += 64) lds.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 30) lds.lens[288 + lane] = 5;
            (void)build_code(lds.lens, 288, lds.lsym, lc, lane);
            (void)build_code(lds.lens + 288, 30, lds.dsym, dc, lane);
        } else {                                                   // dynamic code, 3.2.7
            B.need32();
            const uint32_t hlit = B.take(5) + 257u, hdist = B.take(5) + 1u, hclen = B.take(4) + 4u;
            if (hlit > 286u || hdist > 30u) { err = INF_BAD_LENGTHS; break; }
            if (lane < 19) lds.lens[lane] = 0;
            for (uint32_t k = 0; k < hclen; ++k) { B.need32(); const uint32_t v = B.take(3); if (lane == 0) lds.lens[CLORDER[k]] = (uint8_t)v; }
            Code cc{};
            if (!build_code(lds.lens, 19, lds.csym, cc, lane)) { err = INF_OVERSUBSCRIBED; break; }
            const uint32_t total = hlit + hdist;
            uint32_t k = 0, prev = 0;
            while (k < total && !err) {
                B.need32();
                const uint32_t s = decode_sym(B, cc, lds.csym, lane);
                if (s < 16u) { if (lane == 0) lds.lens[32 + k] = (uint8_t)s; prev = s; ++k; continue; }
                if (s > 18u) { err = INF_BAD_CODE; break; }
                uint32_t rep, val = 0;
                if (s == 16u) { if (k == 0) { err = INF_BAD_LENGTHS; break; } val = prev; rep = 3u + B.take(2); }
                else if (s == 17u) rep = 3u + B.take(3);
                else rep = 11u + B.take(7);
                if (k + rep > total) { err = INF_BAD_LENGTHS; break; }
                if ((uint32_t)lane < rep) lds.lens[32 + k + lane] = (uint8_t)val;
                if ((uint32_t)lane + 64u < rep) lds.lens[32 + k + lane + 64] = (uint8_t)val;
                if ((uint32_t)lane + 128u < rep) lds.lens[32 + k + lane + 128] = (uint8_t)val;
                prev = val; k += rep;
            }
            if (err) break;
            if (rfl(lds.lens[32 + 256]) == 0u) { err = INF_BAD_LENGTHS; break; }      // no end-of-block code
            if (!build_code(lds.lens + 32, (int)hlit, lds.lsym, lc, lane)) { err = INF_OVERSUBSCRIBED; break; }
            if (!build_code(lds.lens + 32 + hlit, (int)hdist, lds.dsym, dc, lane)) { err = INF_OVERSUBSCRIBED; break; }
        }
        fill_lut<LBITS>(lc, lds.lsym, lds.llut, lane);
        fill_lut<DBITS>(dc, lds.dsym, lds.dlut, lane);
        park_code(lc, lds.lcode, lane); park_code(dc, lds.dcode, lane);
        // literal / length + distance symbols until end-of-block
        for (;;) {
            // literals with short codes, the bulk of the symbols, stay in this inner loop: table hit, one byte into
            // the lane register, a 64-byte store when it is full
            uint32_t e, ev;
            for (;;) {
                B.need32();
                ev = lds.llut[(uint32_t)B.bb & ((1u << LBITS) - 1u)];
                e = rfl(ev);
                if (e - 1u >= (256u << 4) - 1u) break;             // not a short-code literal (0: long code; >= 256 << 4: length / end of block)
                const uint32_t l = ev & 15u;
                B.bb >>= l; B.nb -= (int)l;
                if ((uint32_t)lane == npend) lit = ev >> 4;
                if (++npend == 64u) {
                    if (pos + 64u > isize) break;                  // flush() reports the overrun below
                    outp[pos + lane] = (uint8_t)lit;
                    pos += 64u; npend = 0;
                }
            }
            if (npend == 64u) { flush(); break; }
            uint32_t s;
            if (e & 15u) { const uint32_t l = ev & 15u; B.bb >>= l; B.nb -= (int)l; s = e >> 4; }
            else s = decode_sym(B, fetch_code(lds.lcode, lane), lds.lsym, lane);
            if (s < 256u) {                                        // a literal with a long code
                if ((uint32_t)lane == npend) lit = s;
                if (++npend == 64u) { flush(); if (err) break; }
                continue;
            }
            if (s == 256u) break;
            if (s == ~0u) { err = B.bitpos() > end_bit ? INF_OVERRUN_IN : INF_BAD_CODE; break; }
            s -= 257u;
            if (s >= 29u) { err = INF_BAD_CODE; break; }
            // length and distance (RFC 1951 3.2.5 in closed form) on the vector pipe: the values are uniform, but the
            // arithmetic runs in VGPRs (B.vz taints it) and only the two results are read back
            const uint32_t sv = s | B.vz;
            // branch-free: the extra-bit count and the base, selected by masks (a ?: on these "lane-varying" values
            // would be compiled as divergent control flow)
            const uint32_t l_small = 0u - (uint32_t)(sv < 8u), l_top = 0u - (uint32_t)(sv == 28u);
            const uint32_t le = ((sv >> 2) - 1u) & ~(l_small | l_top);
            const uint32_t lbase = ((3u + sv) & l_small) | (258u & l_top) | ((3u + ((4u + (sv & 3u)) << le)) & ~(l_small | l_top));
            const uint32_t lextra = (uint32_t)B.bb & ((1u << le) - 1u);
            B.bb >>= le; B.nb -= (int)le;
            B.need32();
            const uint32_t dev = lds.dlut[(uint32_t)B.bb & ((1u << DBITS) - 1u)];
            const uint32_t du = rfl(dev);
            uint32_t dsv, ds;
            if (du & 15u) { const uint32_t l = dev & 15u; B.bb >>= l; B.nb -= (int)l; dsv = dev >> 4; ds = du >> 4; }
            else { ds = decode_sym(B, fetch_code(lds.dcode, lane), lds.dsym, lane); dsv = ds | B.vz; }
            if (ds >= 30u) { err = INF_BAD_CODE; break; }
            const uint32_t d_small = 0u - (uint32_t)(dsv < 4u);
            const uint32_t de = ((dsv >> 1) - 1u) & ~d_small;
            const uint32_t dbase = ((1u + dsv) & d_small) | ((1u + ((2u + (dsv & 1u)) << de)) & ~d_small);
            const uint32_t dextra = (uint32_t)B.bb & ((1u << de) - 1u);
            B.bb >>= de; B.nb -= (int)de;
            const uint32_t lenv = lbase + lextra, distv = dbase + dextra;
            // the common match (one step, source entirely in bytes already stored): its load is issued BEFORE the
            // pending literals' store, so the wait for the data does not sit behind that store.  The four conditions
            // len <= 64, dist >= len + npend, dist <= pos + npend, pos + npend + len <= isize as one sign test
            // (all operands are below 2^17), evaluated on the vector pipe.
            const uint32_t have = pos + npend;
            const uint32_t neg = (64u - lenv) | (distv - lenv - npend) | (have - distv) | (isize - have - lenv);
            if (__ballot((int32_t)neg >= 0)) {
                uint8_t v = 0;
                if ((uint32_t)lane < lenv) v = outp[have - distv + (uint32_t)lane];
                if ((uint32_t)lane < npend) outp[pos + lane] = (uint8_t)lit;
                if ((uint32_t)lane < lenv) outp[have + (uint32_t)lane] = v;
                pos = have + rfl(lenv); npend = 0;
                continue;
            }
            const uint32_t len = rfl(lenv), dist = rfl(distv);
            flush();
            if (err) break;
            if (dist > pos) { err = INF_BAD_DISTANCE; break; }
            if (pos + len > isize) { err = INF_OVERRUN_OUT; break; }
            const uint8_t* src = outp + pos - dist;
#ifdef BGP_INFLATE_STATS
            if (lane == 0) {
                atomicAdd(&g_stats[0], 1ull); atomicAdd(&g_stats[7], (unsigned long long)len);
                for (int k = 1; k <= 6; ++k) if (dist <= (512u << k)) atomicAdd(&g_stats[k], 1ull);
            }
#endif
            if (dist >= len) {
                for (uint32_t k = lane; k < len; k += 64) outp[pos + k] = src[k];
            } else if (dist == 1) {
                const uint8_t v = src[0];
                for (uint32_t k = lane; k < len; k += 64) outp[pos + k] = v;
            } else {
                for (uint32_t k = lane; k < len; k += 64) outp[pos + k] = src[k % dist];
            }
            pos += len;
            if (B.bitpos() > end_bit) { err = INF_OVERRUN_IN; break; }
        }
    }
    if (!err) flush();
    if (!err && B.bitpos() > end_bit) err = INF_OVERRUN_IN;
    if (!err && pos != isize) err = INF_SHORT_OUT;
    if (!err) {                                                    // crc32 of the output against the footer (block.d:197-207)
        const uint32_t S = (((isize + 63u) >> 6) + 3u) & ~3u;
        const uint32_t beg = min(isize, (uint32_t)lane * S), end = min(isize, beg + S);
        uint32_t c = 0;
        if (beg < end) {
            c = ~0u;
            uint32_t i = beg;
            for (; i + 4 <= end; i += 4) {
                uint32_t w; __builtin_memcpy(&w, outp + i, 4);
                c ^= w;
                c = crc_t[c & 0xFF] ^ (c >> 8); c = crc_t[c & 0xFF] ^ (c >> 8); c = crc_t[c & 0xFF] ^ (c >> 8); c = crc_t[c & 0xFF] ^ (c >> 8);
            }
            for (; i < end; ++i) c = crc_t[(c ^ outp[i]) & 0xFF] ^ (c >> 8);
            c = ~c;
            uint32_t after = isize - end, p = 1u << 31; int k = 3;       // x^(8 * after): zlib x2nmodp(after, 3)
            while (after) { if (after & 1u) p = multmodp_c(CRC.x2n[k & 31], p); after >>= 1; ++k; }
            c = multmodp_c(p, c);
        }
#pragma unroll
        for (int d = 32; d; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d);
        if (c != bk.crc) err = INF_BAD_CRC;
    }
    if (lane == 0) status[bi] = err;
}

}  // namespace

// Inflates n_blk blocks (descriptors on the device) from comp into out; status[i] = 0 or the reason block i failed.
int inflate_blocks_device(bgp_ctx* c, const uint8_t* d_comp, uint8_t* d_out, const bgp_inflate_block* d_blk, uint32_t n_blk, int32_t* d_status, uint64_t out_bytes, hipStream_t on) {
    if (!n_blk) return BGP_OK;
    hipStream_t st = on ? on : c->stream;
    TimedRegion t(c, "k_inflate", out_bytes, st);
    hipLaunchKernelGGL(k_inflate, dim3((n_blk + WAVES - 1) / WAVES), dim3(64 * WAVES), 0, st, d_comp, d_out, d_blk, n_blk, d_status);
    BGP_HIP(hipGetLastError());
#ifdef BGP_INFLATE_STATS
    unsigned long long h[8];
    BGP_HIP(hipStreamSynchronize(c->stream));
    BGP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stats), sizeof h));
    fprintf(stderr, "[biodgpu] inflate stats: %llu matches (%.1f bytes each, %.1f%% of the output); distance <= 1k %.1f%%, 2k %.1f%%, 4k %.1f%%, 8k %.1f%%, 16k %.1f%%, 32k %.1f%%\n",
            h[0], (double)h[7] / (h[0] ? h[0] : 1), 100.0 * h[7] / (out_bytes ? out_bytes : 1), 100.0 * h[1] / h[0], 100.0 * h[2] / h[0], 100.0 * h[3] / h[0], 100.0 * h[4] / h[0], 100.0 * h[5] / h[0], 100.0 * h[6] / h[0]);
#endif
    return BGP_OK;
}
