// Microbenchmark: what does one s_barrier cost a workgroup of 4 / 8 waves, alone and next to MFMA / LDS-read / LDS-DMA work?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/barrier_cost.hip -o tools/ubench/bin/barrier_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, const _Float16* src) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    f4 acc[20];
    for (int i = 0; i < 20; ++i) acc[i] = f4{0, 0, 0, 0};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t grp = wave >> 2;
    h8 fr[9];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { BAR(); }
        if (MODE == 1) {                       // barrier + 20 MFMAs, all waves
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            BAR();
        }
        if (MODE == 2) {                       // ping-pong skeleton: group (it & 1) multiplies, the other one idles; one barrier per slot
            if ((it & 1) == (int)grp) {
#pragma unroll
                for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            }
            BAR();
        }
        if (MODE == 3) {                       // ping-pong: one group multiplies, the other reads 9 fragments from LDS
            if ((it & 1) == (int)grp) {
#pragma unroll
                for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i % 9], b, acc[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) fr[i] = *reinterpret_cast<const h8*>(lds + ((it & 3) * 26624 + i * 1024 + lane * 16));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            BAR();
        }
        if (MODE == 4) {                       // ping-pong: the loading group also issues 3 LDS-DMA pieces (saddr + voffset form)
            if ((it & 1) == (int)grp) {
#pragma unroll
                for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i % 9], b, acc[i], 0, 0, 0);
            } else {
                const _Float16* base = src + (size_t)(it & 63) * 4096;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (wave * 3 + i) * 512 + lane * 8),
                                                     (__attribute__((address_space(3))) void*)(lds + ((it + 3) & 3) * 26624 + (wave * 3 + i) * 1024), 16, 0, 0);
#pragma unroll
                for (int i = 0; i < 9; ++i) fr[i] = *reinterpret_cast<const h8*>(lds + ((it & 3) * 26624 + i * 1024 + lane * 16));
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            BAR();
        }
        if (MODE == 5) {                       // no ping-pong: every wave loads then multiplies, one barrier per step (the 4-wave structure)
#pragma unroll
            for (int i = 0; i < 9; ++i) fr[i] = *reinterpret_cast<const h8*>(lds + ((it & 3) * 26624 + i * 1024 + lane * 16));
#pragma unroll
            for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[i % 9], b, acc[i], 0, 0, 0);
            BAR();
        }
    }
    float s = 0;
    for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads, int iters, float* out, const _Float16* src) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads, 110 * 1024>>>(out, 100, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, threads, 110 * 1024>>>(out, iters, src);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s threads=%3d  %8.1f ns per iteration\n", name, threads, ms * 1e6 / iters);
}

int main() {
    float* out; hipMalloc(&out, 4096);
    _Float16* src; hipMalloc(&src, 64 * 4096 * 2 + 65536); hipMemset(src, 0, 64 * 4096 * 2 + 65536);
    const int it = 20000;
    for (int th : {256, 512}) {
        run<0>("barrier only", th, it, out, src);
        run<1>("20 MFMA (all waves) + barrier", th, it, out, src);
        run<5>("9 ds_read_b128 + 20 MFMA (all waves) + barrier", th, it, out, src);
    }
    run<2>("ping-pong: 20 MFMA | idle, barrier per slot", 512, it, out, src);
    run<3>("ping-pong: 20 MFMA | 9 ds_read_b128, barrier per slot", 512, it, out, src);
    run<4>("ping-pong: 20 MFMA | 3 LDS-DMA + 9 ds_read, barrier per slot", 512, it, out, src);
    return 0;
}
