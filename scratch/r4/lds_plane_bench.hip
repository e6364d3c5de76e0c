// Round 4 microbenchmark: LDS atomic cost of the byte-plane accumulate (lane = descriptor piece, step = chunk of 8 positions;
// 4 classes x even/odd byte-field words per chunk) against today's one-add-per-base pattern.
//   hipcc --offload-arch=gfx950 -O3 scratch/r4/lds_plane_bench.hip -o /tmp/lpb && /tmp/lpb
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int T = 256, PL = 136;              // plane words (128 chunks + pad)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: 8 x ds_add_u32, lanes = consecutive descriptors (3 positions apart)
// MODE 1: 4 x ds_add_u64, same
// MODE 2: 8 x ds_add_u32, stride-3 permuted descriptors (lane L <-> descriptor 3L + r)
// MODE 3: 4 x ds_add_u64, stride-3
// MODE 4: 8 x ds_add_u32, lanes fully distinct consecutive words (ideal)
// MODE 5: 4 x ds_add_u64 ideal
// MODE 6: today's k4_acc: lanes along reads (20 lanes a read), 8 adds, class moves the array
// MODE 7: 8 x ds_add_u32, consecutive descriptors, bank swizzle m + (m >> 4)
template <int MODE> __global__ __launch_bounds__(T) void k(uint32_t* out, int reps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (int i = threadIdx.x; i < 7000; i += T) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; ++r) {
        const uint32_t h = mix(blockIdx.x * 7919u + r * 104729u + wave * 131u);
        uint32_t m0;
        if (MODE == 0 || MODE == 1 || MODE == 7) m0 = ((h % 400u) + lane * 3u + (mix(h + lane) % 3u)) >> 3;
        else if (MODE == 2 || MODE == 3) m0 = ((h % 200u) + lane * 9u + (mix(h + lane) % 9u)) >> 3;
        else m0 = lane;
        if (MODE == 7) m0 = m0 + (m0 >> 4);
        if (MODE == 8 || MODE == 9) m0 = ((h % 400u) + lane * 3u + (mix(h + lane) % 3u)) >> 3;
        const uint32_t seed = mix(h ^ (lane * 2654435761u));
        if (MODE == 6) {
            // 3.2 reads an instruction: lane -> (read, chunk)
            const uint32_t rd = lane / 20, ch = lane % 20;
            const uint32_t pos = (h % 700u) + rd * 5u;
            const uint32_t m = (pos >> 3) + ch;
#pragma unroll 2
            for (int j = 0; j < 11; ++j) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const uint32_t cls = (seed >> (2 * k + (j & 3))) & 3u;
                    atomicAdd(&lds[cls * 4 * PL + (k >> 1) * PL + ((m + j) & 127)], 1u << (16 * (k & 1)));
                }
            }
        } else if (MODE == 8) {          // 4 private copies by lane & 3, a copy 8 banks further
            uint32_t* cp = lds + (lane & 3) * (8 * PL + 8);
#pragma unroll
            for (int j = 0; j < 11; ++j) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t v = (seed >> (c + j)) & 0x01010101u;
                    atomicAdd(&cp[c * PL + ((m0 + j) % PL)], v);
                }
            }
        } else if (MODE == 9) {
            unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds) + (lane & 3) * (4 * PL + 4);
#pragma unroll
            for (int j = 0; j < 11; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned long long v = ((seed >> (c + j)) & 0x01010101u) | ((unsigned long long)((seed >> (c + j + 3)) & 0x01010101u) << 32);
                    atomicAdd(&l64[c * PL + ((m0 + j) % PL)], v);
                }
            }
        } else if (MODE & 1 && MODE != 7) {
            unsigned long long* l64 = reinterpret_cast<unsigned long long*>(lds);
#pragma unroll
            for (int j = 0; j < 11; ++j) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned long long v = ((seed >> (c + j)) & 0x01010101u) | ((unsigned long long)((seed >> (c + j + 3)) & 0x01010101u) << 32);
                    atomicAdd(&l64[c * PL + ((m0 + j) % PL)], v);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 11; ++j) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint32_t v = (seed >> (c + j)) & 0x01010101u;
                    atomicAdd(&lds[c * PL + ((m0 + j) % PL)], v);
                }
            }
        }
    }
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < 7000; i += T) s += lds[i];
    if (s == 0xFFFFFFFFu) out[blockIdx.x] = s;
}

template <int MODE> void run(const char* name, uint32_t* out) {
    const int blocks = 256 * 5 * 4, reps = 64;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 30 * 1024, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), 30 * 1024, 0, out, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double steps = (double)blocks * (T / 64) * reps * 11;          // wave chunk-steps
    printf("%-52s %8.3f ms  %.1f clk per wave chunk-step and CU (2.4 GHz, 256 CU)\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / steps);
}

int main() {
    uint32_t* out; hipMalloc(&out, 1 << 20);
    run<6>("today: lanes along reads, 8 adds by class", out);
    run<0>("8 x u32, consecutive descriptors", out);
    run<1>("4 x u64, consecutive descriptors", out);
    run<2>("8 x u32, stride-3 descriptors", out);
    run<3>("4 x u64, stride-3 descriptors", out);
    run<4>("8 x u32, distinct consecutive words", out);
    run<5>("4 x u64, distinct consecutive words", out);
    run<7>("8 x u32, consecutive descriptors, swizzled", out);
    run<8>("8 x u32, consecutive descriptors, 4 copies by lane", out);
    run<9>("4 x u64, consecutive descriptors, 4 copies by lane", out);
    return 0;
}
