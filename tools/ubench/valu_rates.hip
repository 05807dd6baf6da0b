// VALU issue-rate microbenchmark for gfx950: how many lane-ops per clock per CU for the integer ops Philox needs.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
    uint64_t acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + a[i] * 1e-9f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] ^ (a[i] >> 3);                                                    // 2 plain VALU (shift+xor) baseline
            if (OP == 1) { uint64_t p = (uint64_t)0xD2511F53u * a[i] + acc[i]; acc[i] = p; a[i] = (uint32_t)(p >> 32) ^ (uint32_t)p; }  // mad_u64_u32 + xor
            if (OP == 2) a[i] = a[i] * 0xCD9E8D57u + 1u;                                               // mul_lo (+add / mad)
            if (OP == 3) a[i] = __umulhi(a[i], 0xCD9E8D57u) + a[i];                                    // mul_hi + add
            if (OP == 4) f[i] = fmaf(f[i], 1.0000001f, 1e-7f);                                         // fma f32
            if (OP == 5) f[i] = __log2f(f[i]) + 2.0f;                                                  // log + add
            if (OP == 6) a[i] = __builtin_rotateleft32(a[i], 7) + a[i];                                // alignbit + add
            if (OP == 7) a[i] = (a[i] & 0xFFFFFFu) * 0x5bd1e9u + 1u;                                   // mul_u32_u24-able
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ (uint32_t)acc[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int OP> void run(const char* name, int ops_per_iter) {
    uint32_t* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<OP><<<2048, 256>>>(d, 1); hipDeviceSynchronize();
    hipEventRecord(s); k<OP><<<2048, 256>>>(d, 2); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double laneops = 2048.0 * 256 * ITERS * 8 * ops_per_iter;
    printf("%-28s %8.3f ms  %7.2f Tlane-op/s (counting %d VALU per element-iter)\n", name, ms, laneops / ms / 1e9, ops_per_iter);
    hipFree(d);
}
int main() {
    run<0>("shift+xor", 2); run<1>("mad_u64_u32 + shift/xor", 2); run<2>("mul_lo(+add)", 1); run<3>("mul_hi + add", 2);
    run<4>("fma_f32", 1); run<5>("log2_f32 + add", 2); run<6>("rotl + add", 2); run<7>("and + mul24/mad24", 2);
    return 0;
}
