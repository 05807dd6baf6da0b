// Microbenchmark: LDS-DMA (global_load_lds_dwordx4) throughput per CU for different source patterns, L2-resident source.
//   lanes-per-row 4 (64 B of a row per 4 lanes, 16 rows per instruction), 8 (128 B, 8 rows), 64 (1 KiB contiguous)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_patterns.hip -o tools/ubench/bin/dma_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int LPR, int VGPR_DST>
__global__ __launch_bounds__(512, 2) void k(const uint8_t* src, int iters, int row_stride, float* out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* base = src + (size_t)(blockIdx.x & 31) * (1u << 20);     // 1 MiB window per block: L2 / MALL resident
    constexpr int ROWS = 64 / LPR;
    const uint32_t row = lane / LPR, col = (lane % LPR) * 16u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = ((it * 8 + j) * 8 + wave) * ROWS + row;
            const uint8_t* p = base + ((size_t)r * row_stride + ((it & 1) * LPR * 16) + col) % (1u << 20);
            if (VGPR_DST) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                acc += __uint_as_float(v.x & 0x3f800000u);
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * 8 + j) * 1024), 16, 0, 0);
            }
        }
        if (!VGPR_DST) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 12345.f) out[threadIdx.x] = acc;
}

template <int LPR, int VGPR_DST>
void run(const char* name, const uint8_t* src, int row_stride, float* out) {
    hipFuncSetAttribute((const void*)k<LPR, VGPR_DST>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<LPR, VGPR_DST><<<256, 512, 64 * 1024>>>(src, 50, row_stride, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<LPR, VGPR_DST><<<256, 512, 64 * 1024>>>(src, iters, row_stride, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)iters * 8 * 8 * 1024;
    printf("%-44s row stride %6d B: %7.1f GB/s per CU  (%6.2f TB/s chip, %5.1f ns per 1-KiB instruction per CU)\n", name, row_stride,
           bytes_per_cu / (ms * 1e-3) / 1e9, bytes_per_cu * 256 / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 64.0));
}

int main() {
    uint8_t* src; hipMalloc(&src, 33u << 20); hipMemset(src, 0, 33u << 20);
    float* out; hipMalloc(&out, 4096);
    for (int stride : {640, 2560, 10240}) {
        run<4, 0>("LDS-DMA, 4 lanes x 16 B per row (64-B rows)", src, stride, out);
        run<8, 0>("LDS-DMA, 8 lanes x 16 B per row (128-B rows)", src, stride, out);
        run<64, 0>("LDS-DMA, 64 lanes contiguous (1 KiB)", src, stride, out);
        run<4, 1>("global_load_dwordx4 -> VGPR, 64-B rows", src, stride, out);
        run<8, 1>("global_load_dwordx4 -> VGPR, 128-B rows", src, stride, out);
    }
    return 0;
}
