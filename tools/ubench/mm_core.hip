// Microbenchmark of the matmul engine's multiplying loop in isolation (no DMA, LDS filled once): per phase 9 ds_read_b128 of the NEXT fragments
// + 20 v_mfma_f32_16x16x32_f16 on the current ones, one barrier per two phases -- what the 8 multiplying waves of gsw_mm_kernel do.
//   MODE 0: the kernel's addressing (swizzled 128-byte rows, 4 A + 5 W fragments), two fragment sets, 4 x 5 accumulators
//   MODE 1: same, but every fragment read hits ONE address per lane group (no bank traffic differences)
//   MODE 2: same reads as MODE 0, MFMAs all on the same two operand registers (register-file pressure of operand fetch removed)
//   MODE 3: MODE 0 without the reads (MFMAs + barrier only)
//   NVALU (template): that many extra independent v_fma_f32 per phase, left to the scheduler to place between the MFMAs -- what an epilogue of the
//   previous tile issued inside the next tile's main loop would cost
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mm_core.hip -o tools/ubench/bin/mm_core
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// RANDOM: LDS holds hashed fp16 values in (-1, 1) instead of a near-constant pattern (operand toggling costs power, power costs clock)
template <int MODE, int THREADS, bool RANDOM = false, int NVALU = 0>
__global__ __launch_bounds__(THREADS, THREADS == 512 ? 2 : 1) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr uint32_t STAGE = 416u * 128u, RING = 3u * STAGE;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (uint32_t i = tid; i < RING / 16; i += THREADS) {
        if (RANDOM) {
            uint32_t w[4];
            for (int j = 0; j < 4; ++j) {
                uint32_t h = (i * 4u + j + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
                const _Float16 lo = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)), hi = (_Float16)(((int)(h >> 16) - 32768) * (1.0f / 32768.0f));
                uint16_t a, b; __builtin_memcpy(&a, &lo, 2); __builtin_memcpy(&b, &hi, 2);
                w[j] = (uint32_t)a | ((uint32_t)b << 16);
            }
            reinterpret_cast<uint4*>(lds)[i] = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            reinterpret_cast<uint4*>(lds)[i] = make_uint4(0x3c003c00u + i, 0x38003800u, 0x3c003a00u, 0x34003c00u);
        }
    }
    __syncthreads();
    if (wave >= 8) {                      // SPLIT layout: the four extra waves only meet the barriers
        BAR();
        for (int it = 0; it < iters; ++it) BAR();
        return;
    }
    const uint32_t grp = (wave >> 2) & 1u, wm = wave & 3u;
    const uint32_t lane_rd0 = MODE == 1 ? (lane & 15u) * 16u : (lane & 15u) * 128u + ((((lane >> 4)) ^ ((lane >> 1) & 7u)) << 4);
    const uint32_t a_rd0 = wm * 64u * 128u + lane_rd0, a_rd1 = a_rd0 ^ 64u;
    const uint32_t w_rd0 = 256u * 128u + (grp * 80u) * 128u + lane_rd0, w_rd1 = w_rd0 ^ 64u;
    f4 acc[5][4];
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = f4{0, 0, 0, 0};
    h8 xa[4], wa[5], xb[4], wb[5];
    auto read_frags = [&](h8 (&xf)[4], h8 (&wf)[5], uint32_t slot, uint32_t khalf) {
        if (MODE == 3) return;
        const uint8_t* ap = lds + ((khalf ? a_rd1 : a_rd0) + slot);
        const uint8_t* wp = lds + ((khalf ? w_rd1 : w_rd0) + slot);
#pragma unroll
        for (int im = 0; im < 4; ++im) xf[im] = *reinterpret_cast<const h8*>(ap + (MODE == 1 ? 0 : im * 2048));
#pragma unroll
        for (int in = 0; in < 5; ++in) wf[in] = *reinterpret_cast<const h8*>(wp + (MODE == 1 ? 0 : in * 2048));
    };
    auto mfma20 = [&](h8 (&xc)[4], h8 (&wc)[5]) {
#pragma unroll
        for (int in = 0; in < 5; ++in)
#pragma unroll
            for (int im = 0; im < 4; ++im)
                acc[in][im] = MODE == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[0], xc[0], acc[in][im], 0, 0, 0)
                                        : __builtin_amdgcn_mfma_f32_16x16x32_f16(wc[in], xc[im], acc[in][im], 0, 0, 0);
    };
    auto pin = [&]() {
        if (MODE == 3) return;
#pragma unroll
        for (int i = 0; i < 9; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
    };
    float vx[8];
    for (int i = 0; i < 8; ++i) vx[i] = 1.0f + 0.001f * (float)(lane + i);
    auto valu = [&]() {
#pragma unroll
        for (int i = 0; i < NVALU; ++i) vx[i & 7] = __builtin_fmaf(vx[i & 7], 1.0001f, 0.0003f * (float)(i + 1));
    };
    for (int i = 0; i < 4; ++i) xb[i] = h8{}; for (int i = 0; i < 5; ++i) wb[i] = h8{};
    read_frags(xa, wa, 0u, 0u);
    if (MODE == 3) { for (int i = 0; i < 4; ++i) { xa[i] = *reinterpret_cast<const h8*>(lds + a_rd0 + i * 2048); } for (int i = 0; i < 5; ++i) wa[i] = *reinterpret_cast<const h8*>(lds + w_rd0 + i * 2048);
                     for (int i = 0; i < 4; ++i) xb[i] = xa[i]; for (int i = 0; i < 5; ++i) wb[i] = wa[i]; }
    BAR();
    uint32_t rd_slot = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t nx = rd_slot + STAGE == RING ? 0u : rd_slot + STAGE;
        read_frags(xb, wb, rd_slot, 1u);
        mfma20(xa, wa);
        valu();
        pin();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BAR();
        read_frags(xa, wa, nx, 0u);
        mfma20(xb, wb);
        valu();
        pin();
        __builtin_amdgcn_sched_barrier(0);
        rd_slot = nx;
    }
    float s = 0;
    for (int a = 0; a < 5; ++a) for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][3];
    for (int i = 0; i < 8; ++i) s += vx[i];
    out[blockIdx.x * THREADS + tid] = s;
}

template <int MODE, int THREADS, bool RANDOM = false, int NVALU = 0>
void run(const char* name, float* out, int iters = 2000) {
    hipFuncSetAttribute((const void*)k<MODE, THREADS, RANDOM, NVALU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, THREADS, RANDOM, NVALU>), dim3(256), dim3(THREADS), 3 * 416 * 128, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS, RANDOM, NVALU>), dim3(256), dim3(THREADS), 3 * 416 * 128, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / iters;           // per stage (two phases): 40 MFMAs per wave, 80 per SIMD = 1280 pipe cycles
    printf("%-64s threads=%d  %7.1f ns per stage  -> %5.1f %% of the MFMA rate at 2.4 GHz (%.0f TFLOP/s equivalent)\n", name, THREADS, ns, 1280.0 / 2.4 / ns * 100.0,
           256.0 * 8 * 40 * 16384 / ns / 1e3);
}

int main() {
    float* out; hipMalloc(&out, 256 * 768 * 4);
    run<3, 512>("MFMAs + barrier only", out);
    run<0, 512>("kernel addressing, 2 fragment sets", out);
    run<1, 512>("all reads on one address", out);
    run<2, 512>("kernel reads, MFMAs on two fixed operand registers", out);
    run<0, 768>("kernel addressing, 12 waves (4 only meet the barrier)", out);
    run<3, 768>("MFMAs + barrier only, 12 waves", out);
    run<0, 512, true>("kernel addressing, RANDOM operands", out);
    run<3, 512, true>("MFMAs + barrier only, RANDOM operands", out);
    run<0, 768, true>("kernel addressing, 12 waves, RANDOM operands", out);
    run<0, 512, true, 40>("kernel addressing, RANDOM operands, + 40 VALU per phase", out);
    run<0, 512, true, 80>("kernel addressing, RANDOM operands, + 80 VALU per phase", out);
    run<0, 512, true, 160>("kernel addressing, RANDOM operands, + 160 VALU per phase", out);
    run<0, 512, true>("kernel addressing, RANDOM operands, 100x longer run", out, 200000);
    run<0, 512, false>("kernel addressing, constant operands, 100x longer run", out, 200000);
    return 0;
}
