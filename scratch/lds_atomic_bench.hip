// Microbenchmark: ds_add_u32 throughput for K4's access pattern (lane = read, all lanes step the same query index)
// vs the same adds issued in a per-lane rotated order that makes the 64 lanes of every instruction hit 32 distinct
// banks twice.  hipcc --offload-arch=gfx950 -O3 scratch/lds_atomic_bench.hip -o /tmp/lds_bench && /tmp/lds_bench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int W = 2048, T = 512, WORDS = 5 * W + 64;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE> __global__ __launch_bounds__(T) void k(uint32_t* out, int reps) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < WORDS; i += T) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    for (int r = 0; r < reps; ++r) {
        // 64 consecutive sorted reads: one read per ~5 columns with jitter, like cfg3
        const uint32_t h = mix(blockIdx.x * 7919u + r * 104729u + wave * 131u);
        const uint32_t pos = (h % 1500u) + lane * 5u + (mix(h + lane) % 5u);
        const uint32_t seed = mix(h ^ (lane * 2654435761u));
        if (MODE == 0) {                       // uniform q
#pragma unroll 8
            for (uint32_t q = 0; q < 160; ++q) {
                const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                const uint32_t col = pos + q;
                if (col < W) atomicAdd(&lds[cls * W + col], 1u);
            }
        } else if (MODE == 1) {                // rotated inside blocks of 32 bases: bank = (lane + j) mod 32
            for (uint32_t blk = 0; blk < 160; blk += 32) {
                const uint32_t rot = (lane - pos - blk) & 31u;
#pragma unroll 8
                for (uint32_t j = 0; j < 32; ++j) {
                    const uint32_t q = blk + ((j + rot) & 31u);
                    const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                    const uint32_t col = pos + q;
                    if (col < W) atomicAdd(&lds[cls * W + col], 1u);
                }
            }
        } else if (MODE == 5) {                // rotated inside each dword (8 bases): bank mod 8 = (lane + b) mod 8
            for (uint32_t q0 = 0; q0 < 160; q0 += 8) {
                const uint32_t rot = (lane - pos - q0) & 7u;
#pragma unroll
                for (uint32_t b = 0; b < 8; ++b) {
                    const uint32_t q = q0 + ((b + rot) & 7u);
                    const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                    const uint32_t col = pos + q;
                    if (col < W) atomicAdd(&lds[cls * W + col], 1u);
                }
            }
        } else if (MODE == 3) {                // ideal: lane-consecutive words
#pragma unroll 8
            for (uint32_t q = 0; q < 160; ++q) {
                const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                atomicAdd(&lds[cls * W + lane + 64 * (q & 15)], 1u);
            }
        } else if (MODE == 4) {                // all lanes of a wave on one bank (worst case)
#pragma unroll 8
            for (uint32_t q = 0; q < 160; ++q) {
                const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                atomicAdd(&lds[(cls * W + lane * 32 + (q & 15)) % (5 * W)], 1u);
            }
        } else {                               // no LDS at all: the arithmetic alone
#pragma unroll 8
            for (uint32_t q = 0; q < 160; ++q) {
                const uint32_t cls = (seed >> ((q & 15) * 2)) & 3u;
                acc += cls * W + pos + q;
            }
        }
    }
    __syncthreads();
    uint32_t s = acc;
    for (int i = threadIdx.x; i < WORDS; i += T) s += lds[i];
    if (s == 0xFFFFFFFFu) out[blockIdx.x] = s;
}

template <int MODE> void run(const char* name, uint32_t* out) {
    const int blocks = 256 * 8, reps = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), WORDS * 4 + 30000, 0, out, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(T), WORDS * 4 + 30000, 0, out, reps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double adds = (double)blocks * T * reps * 160;
    printf("%-28s %8.3f ms  %.3e lane-adds/s  %.2f lane-adds/clk/CU (2.4 GHz, 256 CU)\n", name, ms, adds / (ms * 1e-3), adds / (ms * 1e-3) / 2.4e9 / 256);
}

int main() {
    uint32_t* out; hipMalloc(&out, 1 << 20);
    run<2>("arithmetic only", out);
    run<0>("uniform q (K4 today)", out);
    run<1>("rotated, 2 lanes per bank", out);
    run<5>("rotated inside a dword (mod 8)", out);
    run<3>("lane-consecutive words", out);
    run<4>("one bank for all lanes", out);
    return 0;
}
