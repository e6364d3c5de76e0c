// What a plain copy gets on this box, by access pattern (k12_lds moves 69.8 GB in 13.9 ms = 5.0 TB/s: is that the ceiling?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// mode bits: 1 nontemporal stores, 2 nontemporal loads; CH: contiguous bytes a workgroup takes at a time
template <int MODE> __global__ __launch_bounds__(512) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, size_t chunk16) {
    const size_t nchunks = (n16 + chunk16 - 1) / chunk16;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t b = c * chunk16, e = b + chunk16 < n16 ? b + chunk16 : n16;
        for (size_t i = b + threadIdx.x; i < e; i += 4 * blockDim.x) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const size_t j = i + (size_t)k * blockDim.x; if (j < e) v[k] = (MODE & 2) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + j)) : *reinterpret_cast<const u32x4*>(src + j); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const size_t j = i + (size_t)k * blockDim.x; if (j < e) { if (MODE & 1) __builtin_nontemporal_store(v[k], reinterpret_cast<u32x4*>(dst + j)); else *reinterpret_cast<u32x4*>(dst + j) = v[k]; } }
        }
    }
}
int main() {
    const size_t bytes = 16ull << 30, n16 = bytes / 16;
    uint4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mode, int grid, size_t chunk) {
        float best = 1e9;
        for (int r = 0; r < 4; ++r) {
            hipEventRecord(e0);
            switch (mode) {
                case 0: hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(512), 0, 0, a, b, n16, chunk / 16); break;
                case 1: hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(512), 0, 0, a, b, n16, chunk / 16); break;
                case 2: hipLaunchKernelGGL(k_copy<2>, dim3(grid), dim3(512), 0, 0, a, b, n16, chunk / 16); break;
                default: hipLaunchKernelGGL(k_copy<3>, dim3(grid), dim3(512), 0, 0, a, b, n16, chunk / 16); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("mode %d (nt stores %d, nt loads %d) grid %6d chunk %7zu B: %.2f ms  %.2f TB/s moved\n", mode, mode & 1, (mode >> 1) & 1, grid, chunk, best, 2.0 * bytes / best / 1e9);
    };
    for (int mode = 0; mode < 4; ++mode)
        for (int grid : {1024, 2048, 8192, 65536})
            for (size_t chunk : {(size_t)32768, (size_t)262144})
                run(mode, grid, chunk);
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
    hipEventRecord(e0); hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemcpy D2D: %.2f ms %.2f TB/s moved\n", ms, 2.0 * bytes / ms / 1e9);
    return 0;
}
