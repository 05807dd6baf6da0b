// LDS fragment bytes per MFMA under the board's power limit: the multiplying loop of a GEMM tile in isolation (LDS filled once with random fp16, no DMA, no barrier; 8 waves
// per workgroup, one workgroup per CU, seconds per variant so that the power management settles), fragments of the NEXT k-step read while the current one multiplies.
//   A  16x16x32, wave tile  64 x 80  (the engine's 256 x 160 tile):  9 fragments per 20 MFMAs, 0.055 LDS bytes per MAC
//   B  16x16x32, wave tile 128 x 80  (the 256 x 320 tile of round 5): 13 per 40,                0.040
//   C  32x32x16, wave tile 128 x 64  (a 256 x 256 tile, 128 accumulator registers): 6 per 8,   0.023
//   D  32x32x16, wave tile 128 x 96  (a 256 x 384 tile, 192 accumulator registers): 7 per 12,  0.018
//   E / F  the MFMAs of B / D alone (no fragment reads)
// What a tile built on the 32 x 32 instruction would be worth at the power limit is C and D against B.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/frag_per_mfma.hip -o tools/ubench/bin/frag_per_mfma ; usage: frag_per_mfma [seconds per variant]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr uint32_t BYTES = 144u * 1024u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (uint32_t i = tid; i < BYTES / 16; i += 512) {
        uint32_t w[4];
        for (int j = 0; j < 4; ++j) {
            uint32_t h = (i * 4u + j + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const _Float16 lo = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)), hi = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
            uint16_t a, b; __builtin_memcpy(&a, &lo, 2); __builtin_memcpy(&b, &hi, 2);
            w[j] = (uint32_t)a | ((uint32_t)b << 16);
        }
        reinterpret_cast<uint4*>(lds)[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    constexpr bool M32 = V == 2 || V == 3 || V == 5;                  // 32x32x16 variants
    constexpr int NA = V == 0 ? 4 : (M32 ? 4 : 8), NW = V == 0 ? 5 : (V == 2 ? 2 : (V == 3 || V == 5 ? 3 : 5));
    constexpr bool READS = V < 4;
    // 128-byte rows, 16-byte chunks XOR-swizzled with (row >> 1) & 7 (the engine's layout); a 32-row fragment takes rows lane & 31, chunk lane >> 5
    const uint32_t row = M32 ? (lane & 31u) : (lane & 15u), ch = M32 ? (lane >> 5) : (lane >> 4);
    const uint32_t a_base = (wave & 1u) * (uint32_t)(NA * (M32 ? 32 : 16)) * 128u + row * 128u;
    const uint32_t w_base = 64u * 1024u + (wave >> 1) * (uint32_t)(NW * (M32 ? 32 : 16)) * 128u + row * 128u;
    // per k-step: the step's NW weight fragments, then the NA activation fragments one ahead of the MFMA group that uses them (the engine streams them the same way)
    auto frag_at = [&](uint32_t base, int i, uint32_t kk) -> h8 {
        const uint32_t c = (((ch + (M32 ? 2u : 4u) * (kk & 1u)) ^ ((row >> 1) & 7u)) & 7u) << 4;
        return *reinterpret_cast<const h8*>(lds + base + (uint32_t)i * (M32 ? 4096u : 2048u) + c);
    };
    h8 wc[NW], wn[NW], x0, x1;
#pragma unroll
    for (int i = 0; i < NW; ++i) wc[i] = frag_at(w_base, i, 0);
    x0 = frag_at(a_base, 0, 0);
    if constexpr (!M32) {
        f4 acc[NW][NA];
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int b = 0; b < NA; ++b) {
                if (READS) { x1 = frag_at(a_base, (b + 1) % NA, (uint32_t)(it + (b + 1) / NA)); if (b < NW) wn[b] = frag_at(w_base, b, (uint32_t)it + 1u); }
#pragma unroll
                for (int a = 0; a < NW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[a], x0, acc[a][b], 0, 0, 0);
                if (READS) x0 = x1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (READS) {
#pragma unroll
                for (int i = 0; i < NW; ++i) wc[i] = wn[i];
            }
        }
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) t += acc[a][b][0] + acc[a][b][3];
        out[blockIdx.x * 512 + tid] = t;
    } else {
        f16v acc[NW][NA];
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b)
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int b = 0; b < NA; ++b) {
                if (READS) { x1 = frag_at(a_base, (b + 1) % NA, (uint32_t)(it + (b + 1) / NA)); if (b < NW) wn[b] = frag_at(w_base, b, (uint32_t)it + 1u); }
#pragma unroll
                for (int a = 0; a < NW; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc[a], x0, acc[a][b], 0, 0, 0);
                if (READS) x0 = x1;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (READS) {
#pragma unroll
                for (int i = 0; i < NW; ++i) wc[i] = wn[i];
            }
        }
        float t = 0.f;
#pragma unroll
        for (int a = 0; a < NW; ++a)
#pragma unroll
            for (int b = 0; b < NA; ++b) t += acc[a][b][0] + acc[a][b][15];
        out[blockIdx.x * 512 + tid] = t;
    }
}
template <int V> static void run(const char* name, float* out, double secs, double macs_per_wave_iter) {
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 144 * 1024, 0, out, iters);
    hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 144 * 1024, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * secs * 1e3 / ms);                   // one long launch: the clock settles under the power limit
    hipEventRecord(e0); hipLaunchKernelGGL(k<V>, dim3(256), dim3(512), 144 * 1024, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * macs_per_wave_iter * iters * 8 * 256;
    printf("%-58s %7.0f TFLOP/s  (%.2f s)\n", name, flops / (ms * 1e-3) / 1e12, ms * 1e-3);
}
int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    float* out; hipMalloc(&out, 256 * 512 * 4);
    run<0>("A 16x16x32  64 x 80 per wave,  9 fragments / 20 MFMAs", out, secs, 20.0 * 16 * 16 * 32);
    run<1>("B 16x16x32 128 x 80 per wave, 13 fragments / 40 MFMAs", out, secs, 40.0 * 16 * 16 * 32);
    run<2>("C 32x32x16 128 x 64 per wave,  6 fragments /  8 MFMAs", out, secs, 8.0 * 32 * 32 * 16);
    run<3>("D 32x32x16 128 x 96 per wave,  7 fragments / 12 MFMAs", out, secs, 12.0 * 32 * 32 * 16);
    run<4>("E 16x16x32 128 x 80 per wave, no fragment reads", out, secs, 40.0 * 16 * 16 * 32);
    run<5>("F 32x32x16 128 x 96 per wave, no fragment reads", out, secs, 12.0 * 32 * 32 * 16);
    return 0;
}
