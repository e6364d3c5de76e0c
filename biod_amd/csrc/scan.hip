// scan.hip -- device-wide prefix scans over uint64 (sum, max) used by framing, decode and pileup.
// Three passes (block reduce -> scan of block totals -> block scan with carry-in); wave64 shuffles
// inside a wave, LDS only for the 4 wave totals of a 256-thread block.
#include "common.h"
#include <algorithm>

namespace {

constexpr int T = 256;
constexpr int ITEMS = 8;
constexpr int TILE = T * ITEMS;

struct OpSum { __device__ static uint64_t id() { return 0; } __device__ static uint64_t f(uint64_t a, uint64_t b) { return a + b; } };
struct OpMax { __device__ static uint64_t id() { return 0; } __device__ static uint64_t f(uint64_t a, uint64_t b) { return a > b ? a : b; } };

template <class Op> __device__ __forceinline__ uint64_t wave_incl(uint64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v = Op::f(o, v);
    }
    return v;
}

// exclusive prefix of per-thread totals across the block; also returns the block total
template <class Op> __device__ __forceinline__ uint64_t block_excl(uint64_t mine, uint64_t* total, uint64_t* lds /*4*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t inc = wave_incl<Op>(mine, lane);
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    uint64_t carry = Op::id();
#pragma unroll
    for (int k = 0; k < T / 64; ++k) if (k < w) carry = Op::f(carry, lds[k]);
    uint64_t tot = Op::id();
#pragma unroll
    for (int k = 0; k < T / 64; ++k) tot = Op::f(tot, lds[k]);
    *total = tot;
    uint64_t prev = __shfl_up(inc, 1, 64);
    uint64_t ex = lane == 0 ? Op::id() : prev;
    __syncthreads();
    return Op::f(carry, ex);
}

template <class Op> __global__ __launch_bounds__(T) void k_reduce(const uint64_t* __restrict__ in, uint64_t* __restrict__ bsum, uint64_t n) {
    __shared__ uint64_t lds[T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * ITEMS;
    uint64_t acc = Op::id();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (base + k < n) acc = Op::f(acc, in[base + k]);
    uint64_t tot;
    (void)block_excl<Op>(acc, &tot, lds);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// single block: bsum[i] <- exclusive prefix; bsum[nb] <- grand total
template <class Op> __global__ __launch_bounds__(T) void k_scan_bsum(uint64_t* bsum, uint64_t nb) {
    __shared__ uint64_t lds[T / 64];
    uint64_t carry = Op::id();
    for (uint64_t base = 0; base < nb; base += T) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < nb ? bsum[i] : Op::id();
        uint64_t tot;
        const uint64_t ex = block_excl<Op>(v, &tot, lds);
        if (i < nb) bsum[i] = Op::f(carry, ex);
        carry = Op::f(carry, tot);
    }
    if (threadIdx.x == 0) bsum[nb] = carry;
}

template <class Op, bool EXCL> __global__ __launch_bounds__(T) void k_down(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                           const uint64_t* __restrict__ bsum, uint64_t n, uint64_t nb) {
    __shared__ uint64_t lds[T / 64];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * ITEMS;
    uint64_t v[ITEMS];
    uint64_t acc = Op::id();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { v[k] = base + k < n ? in[base + k] : Op::id(); acc = Op::f(acc, v[k]); }
    uint64_t tot;
    uint64_t run = Op::f(bsum[blockIdx.x], block_excl<Op>(acc, &tot, lds));
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        if (base + k < n) {
            if (EXCL) { out[base + k] = run; run = Op::f(run, v[k]); }
            else { run = Op::f(run, v[k]); out[base + k] = run; }
        }
    }
    if (EXCL && blockIdx.x == 0 && threadIdx.x == 0) out[n] = bsum[nb];
}

template <class Op, bool EXCL> int run_scan(bgp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* total_host) {
    const uint64_t nb = (n + TILE - 1) / TILE;
    BGP_TRY(c->d_scan_tmp.reserve((nb + 2) * sizeof(uint64_t)));
    uint64_t* bsum = c->d_scan_tmp.as<uint64_t>();
    if (n == 0) {
        if (EXCL) BGP_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t), c->stream));
        if (total_host) *total_host = 0;
        return BGP_OK;
    }
    hipLaunchKernelGGL((k_reduce<Op>), dim3((unsigned)nb), dim3(T), 0, c->stream, in, bsum, n);
    hipLaunchKernelGGL((k_scan_bsum<Op>), dim3(1), dim3(T), 0, c->stream, bsum, nb);
    hipLaunchKernelGGL((k_down<Op, EXCL>), dim3((unsigned)nb), dim3(T), 0, c->stream, in, out, bsum, n, nb);
    BGP_HIP(hipGetLastError());
    if (total_host) {
        BGP_HIP(hipMemcpyAsync(total_host, bsum + nb, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
    }
    return BGP_OK;
}

// three exclusive sum scans of the same length in the same three launches (blockIdx.y picks the array): K0's records / bases /
// CIGAR ops in front of every segment, the long reads' checkpoints / columns / chunks
struct Scan3 { const uint64_t* in[3]; uint64_t* out[3]; uint64_t* bsum; uint64_t n, nb; };
__global__ __launch_bounds__(T) void k_reduce3(Scan3 a) {
    __shared__ uint64_t lds[T / 64];
    const uint64_t* in = a.in[blockIdx.y];
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * ITEMS;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (base + k < a.n) acc += in[base + k];
    uint64_t tot;
    (void)block_excl<OpSum>(acc, &tot, lds);
    if (threadIdx.x == 0) a.bsum[blockIdx.y * (a.nb + 2) + blockIdx.x] = tot;
}
__global__ __launch_bounds__(T) void k_scan_bsum3(Scan3 a) {
    __shared__ uint64_t lds[T / 64];
    uint64_t* bsum = a.bsum + blockIdx.x * (a.nb + 2);
    uint64_t carry = 0;
    for (uint64_t base = 0; base < a.nb; base += T) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < a.nb ? bsum[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_excl<OpSum>(v, &tot, lds);
        if (i < a.nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) bsum[a.nb] = carry;
}
__global__ __launch_bounds__(T) void k_down3(Scan3 a) {
    __shared__ uint64_t lds[T / 64];
    const uint64_t* in = a.in[blockIdx.y]; uint64_t* out = a.out[blockIdx.y];
    const uint64_t* bsum = a.bsum + blockIdx.y * (a.nb + 2);
    const uint64_t base = (uint64_t)blockIdx.x * TILE + (uint64_t)threadIdx.x * ITEMS;
    uint64_t v[ITEMS];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) { v[k] = base + k < a.n ? in[base + k] : 0; acc += v[k]; }
    uint64_t tot;
    uint64_t run = bsum[blockIdx.x] + block_excl<OpSum>(acc, &tot, lds);
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) if (base + k < a.n) { out[base + k] = run; run += v[k]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[a.n] = bsum[a.nb];
}

// a few values from device memory into pinned host memory with ONE small kernel: a hipMemcpyAsync into pageable memory (a
// variable on the stack) is a host round trip each (~20 us), and a step of the path reads a dozen such scalars
struct MailArgs { const uint8_t* src[8]; uint32_t off[8], n[8]; uint8_t* box; int items; };
__global__ __launch_bounds__(256) void k_mail(MailArgs a) {
    for (int k = 0; k < a.items; ++k)
        for (uint32_t b = threadIdx.x; b < a.n[k]; b += 256) a.box[a.off[k] + b] = a.src[k][b];
}

// several buffers filled by ONE launch (blockIdx.y = the buffer): a stage of the path zeroes half a dozen small arrays before its first
// kernel, each a launch of its own through hipMemsetAsync -- a third of a step's ~45 launches, which is what a small stream's step is
struct FillArgs { uint8_t* p[8]; unsigned long long n[8]; uint32_t v[8]; };
__global__ __launch_bounds__(256) void k_fill_many(FillArgs a) {
    uint8_t* p = a.p[blockIdx.y];
    const unsigned long long n = a.n[blockIdx.y];
    const uint32_t w = a.v[blockIdx.y] * 0x01010101u;
    const unsigned long long n16 = n >> 4;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256) reinterpret_cast<uint4*>(p)[i] = make_uint4(w, w, w, w);
    if (blockIdx.x == 0 && threadIdx.x < (n & 15)) p[(n16 << 4) + threadIdx.x] = (uint8_t)w;
}

}  // namespace

int fill_many(bgp_ctx* c, const FillItem* items, int n_items) {
    if (n_items <= 0) return BGP_OK;
    if (n_items > 8) return BGP_EINVAL;
    FillArgs a;
    unsigned long long most = 0;
    for (int k = 0; k < n_items; ++k) {
        if (((uintptr_t)items[k].p & 15) != 0) return BGP_EINVAL;      // (device buffers are 256-byte aligned; a pointer into one must keep 16)
        a.p[k] = (uint8_t*)items[k].p; a.n[k] = items[k].bytes; a.v[k] = items[k].value; most = std::max<unsigned long long>(most, items[k].bytes);
    }
    const unsigned gx = (unsigned)std::min<unsigned long long>(2048, std::max<unsigned long long>(1, (most / 16 + 255) / 256));
    hipLaunchKernelGGL(k_fill_many, dim3(gx, (unsigned)n_items), dim3(256), 0, c->stream, a);
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}

int scan_exclusive_u64x3(bgp_ctx* c, const uint64_t* const in[3], uint64_t* const out[3], uint64_t n, uint64_t* totals_host) {
    const uint64_t nb = (n + TILE - 1) / TILE;
    BGP_TRY(c->d_scan_tmp.reserve(3 * (nb + 2) * sizeof(uint64_t)));
    if (n == 0) {
        for (int k = 0; k < 3; ++k) { BGP_HIP(hipMemsetAsync(out[k], 0, sizeof(uint64_t), c->stream)); if (totals_host) totals_host[k] = 0; }
        return BGP_OK;
    }
    Scan3 a;
    for (int k = 0; k < 3; ++k) { a.in[k] = in[k]; a.out[k] = out[k]; }
    a.bsum = c->d_scan_tmp.as<uint64_t>(); a.n = n; a.nb = nb;
    hipLaunchKernelGGL(k_reduce3, dim3((unsigned)nb, 3), dim3(T), 0, c->stream, a);
    hipLaunchKernelGGL(k_scan_bsum3, dim3(3), dim3(T), 0, c->stream, a);
    hipLaunchKernelGGL(k_down3, dim3((unsigned)nb, 3), dim3(T), 0, c->stream, a);
    BGP_HIP(hipGetLastError());
    if (totals_host) {
        MailItem it[3];
        for (int k = 0; k < 3; ++k) it[k] = MailItem{a.bsum + k * (nb + 2) + nb, totals_host + k, 8};
        BGP_TRY(mail_fetch(c, it, 3));
    }
    return BGP_OK;
}

int mail_fetch(bgp_ctx* c, const MailItem* items, int n_items) {
    if (n_items < 0 || n_items > 8) return BGP_EINVAL;
    BGP_TRY(c->h_mail.reserve(8192));
    MailArgs a;
    a.box = c->h_mail.as<uint8_t>(); a.items = n_items;
    uint32_t off = 0;
    for (int k = 0; k < n_items; ++k) {
        a.src[k] = (const uint8_t*)items[k].src; a.off[k] = off; a.n[k] = items[k].n;
        off += (items[k].n + 15u) & ~15u;
    }
    if (off > 8192) return BGP_EINVAL;
    if (n_items) hipLaunchKernelGGL(k_mail, dim3(1), dim3(256), 0, c->stream, a);
    BGP_HIP(hipGetLastError());
    BGP_HIP(hipStreamSynchronize(c->stream));
    for (int k = 0; k < n_items; ++k) memcpy(items[k].dst, a.box + a.off[k], items[k].n);
    return BGP_OK;
}

int scan_exclusive_u64(bgp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n, uint64_t* total_host) {
    return run_scan<OpSum, true>(c, in, out, n, total_host);
}
int scan_inclusive_max_u64(bgp_ctx* c, const uint64_t* in, uint64_t* out, uint64_t n) {
    return run_scan<OpMax, false>(c, in, out, n, nullptr);
}

// ---- compact transport of a window's columns (bgp_set_compact_columns): 32 bytes a position -> 8, one byte a count.  Most of a window's
// 32 bytes are zeros -- coverage 50 is one byte -- and the windows going back to the host were the critical path of both end-to-end
// paths (1.6 GB for 2.9 GB of records).  A count beyond 255 anywhere raises the flag behind the last position: the caller copies that
// window again in the 32-byte layout.
namespace {
__global__ __launch_bounds__(256) void k_pack_columns8(const uint4* __restrict__ in, uint64_t n, uint2* __restrict__ out, unsigned long long* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 lo = in[2 * i], hi = in[2 * i + 1];
    const uint32_t any = lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w;
    out[i] = make_uint2(lo.x | (lo.y << 8) | (lo.z << 16) | (lo.w << 24), hi.x | (hi.y << 8) | (hi.z << 16) | (hi.w << 24));
    if (any > 255u) atomicOr(flag, 1ull);
}
}  // namespace
int pack_columns8(bgp_ctx* c, const bgp_column* in, uint64_t n, uint8_t* out, hipStream_t st) {
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(out + n * 8);
    BGP_HIP(hipMemsetAsync(flag, 0, 8, st));
    if (!n) return BGP_OK;
    TimedRegion t(c, "k_pack_columns8", n * 40, st);
    hipLaunchKernelGGL(k_pack_columns8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(in), n, reinterpret_cast<uint2*>(out), flag);
    BGP_HIP(hipGetLastError());
    return BGP_OK;
}
