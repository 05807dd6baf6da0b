// Microbenchmark: do MFMA and VALU work of ONE wave overlap on gfx950 when the two streams are independent and interleaved in the instruction
// stream (the premise of software-pipelining the attention loop across tiles: S^T(t + 1) MFMAs under the softmax arithmetic of tile t)?
// Per iteration: NM v_mfma_f32_32x32x16_f16 (independent accumulators) and / or NV packs of softmax-like VALU work (fma, v_exp_f32, add, max, cvt_pk),
// either back to back (MFMAs first, then VALU) or interleaved by sched_group_barrier (1 MFMA, then NV / NM VALU packs).  1, 2 or 3 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int NM = 8;        // MFMAs per iteration (one 64-key tile of S^T at 32 queries per wave)
constexpr int NV = 32;       // softmax elements per lane per iteration

template <int MODE>          // 0: MFMA only; 1: VALU only; 2: MFMA then VALU (source order, no hints); 3: interleaved with sched_group_barrier; 4: VALU without the exponentials
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f16v acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + seed); b[i] = (_Float16)(i * 0.01f + seed); }
    float x[NV];
    for (int i = 0; i < NV; ++i) x[i] = seed * (float)(i + 1) - (float)threadIdx.x * 0.01f;
    float m = seed, rs = 0.f;
    uint32_t packed = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < NM; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
        }
        if (MODE != 0) {
            float mx = x[0];
#pragma unroll
            for (int i = 1; i + 1 < NV; i += 2) mx = fmaxf(fmaxf(mx, x[i]), x[i + 1]);
            m = fmaxf(m, mx * 0.5f);
#pragma unroll
            for (int i = 0; i < NV; i += 2) {
                float e0 = fmaf(x[i], 0.5f, -m), e1 = fmaf(x[i + 1], 0.5f, -m);
                if (MODE != 4) { e0 = __builtin_amdgcn_exp2f(e0); e1 = __builtin_amdgcn_exp2f(e1); }
                rs += e0;
                rs += e1;
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                packed ^= __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{e0, e1}, h2));
                x[i] = e0 * 0.999f + 1e-3f;          // feed back: the next iteration depends on this one (no hoisting), values stay bounded
                x[i + 1] = e1 * 0.999f - 1e-3f;
            }
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);      // then a share of the VALU work (~190 VALU instructions per iteration)
            }
        }
    }
    float s = rs + m + (float)packed;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static double run(int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.25f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    const int iters = 20000;
    const char* names[5] = {"MFMA only (8 x 32x32x16 per iteration)", "VALU only (32 softmax elements per lane)", "MFMA then VALU, compiler's order", "MFMA / VALU interleaved (sched_group_barrier)",
                            "VALU only, exponentials replaced by nothing"};
    for (int wps = 1; wps <= 3; ++wps) {                 // 256-thread blocks: one wave per SIMD each; wps blocks per CU
        const int blocks = 256 * wps;
        double t[5] = {run<0>(blocks, iters, out), run<1>(blocks, iters, out), run<2>(blocks, iters, out), run<3>(blocks, iters, out), run<4>(blocks, iters, out)};
        printf("%d wave(s) per SIMD:\n", wps);
        for (int i = 0; i < 5; ++i) printf("  %-50s %8.3f ms  = %7.1f ns per iteration and wave slot\n", names[i], t[i], t[i] * 1e6 / iters);
        printf("  -> both together cost %.2f x (MFMA + VALU) compiler's order, %.2f x interleaved; 1.00 = no overlap, %.2f = perfect overlap\n", t[2] / (t[0] + t[1]), t[3] / (t[0] + t[1]),
               (t[0] > t[1] ? t[0] : t[1]) / (t[0] + t[1]));
    }
    return 0;
}
