// chase_probe.hip -- what bounds K0's chase?  One lane per segment hops from "record" to "record" (a dependent 4-byte load
// per hop) through a buffer the size of cfg3's stream.  Variants isolate: the side store, the second header load, the
// load flavour, the segment size, record alignment.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/chase_probe.bin scratch/chase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(uint8_t* s, uint64_t len, uint32_t rec) {          // a "block_size" at every multiple of rec
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t off = i * rec;
    if (off + 36 <= len) { uint32_t v[9] = {rec - 4, 0, 0, 11u, 3u, 150u, 0, 0, 0}; __builtin_memcpy(s + off, v, 36); }
}

template <int MODE> __global__ __launch_bounds__(256) void k_chase(const uint8_t* s, uint64_t len, uint32_t seg, uint64_t n_seg, uint32_t rec, uint32_t* side, uint32_t side_cap, uint64_t* out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const uint64_t seg0 = k * (uint64_t)seg, seg_end = min(seg0 + seg, len);
    uint64_t off = (seg0 + rec - 1) / rec * rec;
    uint64_t cnt = 0, ls = 0;
    uint4 sb = make_uint4(0, 0, 0, 0);
    while (off < seg_end && len - off >= 36) {
        uint32_t bs;
        if (MODE & 8) bs = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(s + off));
        else __builtin_memcpy(&bs, s + off, 4);
        if (MODE & 1) { uint32_t f3[3]; __builtin_memcpy(f3, s + off + 12, 12); ls += f3[2] + (f3[1] & 0xFFFF) + (f3[0] & 0xFF); }
        const uint32_t v = (uint32_t)(off - seg0);
        if (MODE & 2) { if (cnt < side_cap) side[k * side_cap + cnt] = v; }
        if (MODE & 4) {
            switch (cnt & 3) { case 0: sb.x = v; break; case 1: sb.y = v; break; case 2: sb.z = v; break; default: sb.w = v; *reinterpret_cast<uint4*>(side + k * side_cap + (cnt & ~3ull)) = sb; }
        }
        ++cnt;
        off += 4 + (uint64_t)bs;
    }
    if ((MODE & 4) && (cnt & 3)) *reinterpret_cast<uint4*>(side + k * side_cap + (cnt & ~3ull)) = sb;
    out[k] = cnt + ls;
}

template <int MODE> float run(const uint8_t* s, uint64_t len, uint32_t seg, uint32_t rec, uint32_t* side, uint64_t* out, int reps) {
    const uint64_t n_seg = (len + seg - 1) / seg;
    const uint32_t side_cap = ((seg / 36 + 2) + 3) & ~3u;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_chase<MODE>, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, 0, s, len, seg, n_seg, rec, side, side_cap, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

// 16 offsets in registers (the iteration count is the same for every active lane: static register indices), one 64-byte line per flush
template <int NB, int ST = 1> __global__ __launch_bounds__(256) void k_chase_buf(const uint8_t* s, uint64_t len, uint32_t seg, uint64_t n_seg, uint32_t rec, uint32_t* side, uint32_t side_cap, uint64_t* out) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_seg) return;
    const uint64_t seg0 = k * (uint64_t)seg, seg_end = min(seg0 + seg, len);
    uint64_t off = (seg0 + rec - 1) / rec * rec;
    uint64_t cnt = 0, ls = 0;
    uint32_t* dst = side + k * side_cap;
    bool live = off < seg_end && len - off >= 36;
    while (live) {
        uint32_t buf[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            buf[e] = 0;
            if (live) {
                uint32_t bs; __builtin_memcpy(&bs, s + off, 4);
                uint32_t f3[3]; __builtin_memcpy(f3, s + off + 12, 12); ls += f3[2] + (f3[1] & 0xFFFF) + (f3[0] & 0xFF);
                buf[e] = (uint32_t)(off - seg0);
                ++cnt;
                off += 4 + (uint64_t)bs;
                live = off < seg_end && len - off >= 36;
            }
        }
#pragma unroll
        for (int q = 0; q < NB / 4; ++q) {
            if (ST == 1) reinterpret_cast<uint4*>(dst)[q] = make_uint4(buf[4 * q], buf[4 * q + 1], buf[4 * q + 2], buf[4 * q + 3]);
            if (ST == 2) { __builtin_nontemporal_store(buf[4 * q], dst + 4 * q); __builtin_nontemporal_store(buf[4 * q + 1], dst + 4 * q + 1); __builtin_nontemporal_store(buf[4 * q + 2], dst + 4 * q + 2); __builtin_nontemporal_store(buf[4 * q + 3], dst + 4 * q + 3); }
            if (ST == 0) ls += buf[4 * q] ^ buf[4 * q + 1] ^ buf[4 * q + 2] ^ buf[4 * q + 3];
        }
        dst += NB;
    }
    out[k] = cnt + ls;
}
template <int NB, int ST = 1> float run_buf(const uint8_t* s, uint64_t len, uint32_t seg, uint32_t rec, uint32_t* side, uint64_t* out, int reps) {
    const uint64_t n_seg = (len + seg - 1) / seg;
    const uint32_t side_cap = ((seg / 36 + 2) + 63) & ~31u;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_chase_buf<NB, ST>), dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, 0, s, len, seg, n_seg, rec, side, side_cap, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    const uint64_t len = argc > 1 ? strtoull(argv[1], 0, 10) : 29000000000ull;
    uint8_t* s; CK(hipMalloc(&s, len + 64));
    uint64_t* out; CK(hipMalloc(&out, ((len >> 12) + 2) * 8));
    uint32_t* side; CK(hipMalloc(&side, (len / 36 + (len >> 10) * 80 + 1024) * 4));
    CK(hipMemset(s, 0, len + 64));
    const uint32_t recs[] = {290};
    const uint32_t segs[] = {65536};
    for (uint32_t rec : recs) {
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((len / rec + 256) / 256)), dim3(256), 0, 0, s, len, rec);
        CK(hipDeviceSynchronize());
        for (uint32_t seg : segs) {
            if (rec != 290 && seg != 65536) continue;
            const double n_rec = (double)len / rec;
            const float t0 = run<0>(s, len, seg, rec, side, out, 3), t1 = run<1>(s, len, seg, rec, side, out, 3), t3 = run<3>(s, len, seg, rec, side, out, 3),
                        t5 = run<5>(s, len, seg, rec, side, out, 3), t9 = run<9>(s, len, seg, rec, side, out, 3), t2 = run<2>(s, len, seg, rec, side, out, 3);
            printf("rec %3u B seg %6u: bs only %.2f ms (%.1f G hops/s) | +hdr12 %.2f | +hdr12 +side4 %.2f | +hdr12 +side16 %.2f | nt +hdr12 %.2f | bs +side4 %.2f\n",
                   rec, seg, t0, n_rec / t0 / 1e6, t1, t3, t5, t9, t2);
            printf("                     +hdr12 + 8 in registers (32 B) %.2f | 16 (64 B) %.2f | 32 (128 B) %.2f\n", run_buf<8>(s, len, seg, rec, side, out, 3), run_buf<16>(s, len, seg, rec, side, out, 3), run_buf<32>(s, len, seg, rec, side, out, 3));
            printf("                     16 in registers: no store %.2f | nontemporal %.2f\n", run_buf<16, 0>(s, len, seg, rec, side, out, 3), run_buf<16, 2>(s, len, seg, rec, side, out, 3));
            fflush(stdout);
        }
    }
    return 0;
}
