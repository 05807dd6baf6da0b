// One "step" of the one-launch cross-attention kernel (csrc/gswm_xattn.hip) in isolation, ONE wave per SIMD (a 512-register kernel): 10 v_mfma_f32_32x32x16 on ten
// accumulators + the other things a step does, switched on one at a time -- what does a step cost when only this wave's own instruction stream can overlap the matrix pipe?
//   bit 0: ten ds_read_b128 fragment reads (each into the registers the previous MFMA consumed)      bit 1: three LDS writes (2 x b128 + b64)
//   bit 2: three global loads (2 x dwordx4 + dwordx2, L2-resident stream)                           bit 3: s_barrier per step
//   bit 4: MFMAs off
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/xattn_step.hip -o xattn_step ; run: ./xattn_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <type_traits>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MASK>
__global__ __launch_bounds__(256) void step_kernel(const uint8_t* __restrict__ stream, float* out, uint64_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[9 * 10240];      // 90 KiB: one workgroup per CU, i.e. ONE wave per SIMD like the real kernel
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t o0 = wave * 1024u + lane * 16u, o1 = o0 + 4096u, o2 = (8u + (wave >> 1)) * 1024u + (wave & 1u) * 512u + lane * 8u;
    for (uint32_t i = threadIdx.x; i < 3 * 10240 / 4; i += 256) reinterpret_cast<uint32_t*>(ring)[i] = 0x3C003C00u;
    __syncthreads();
    f16v acc[10];
    h8 fr[10], xb;
    for (int i = 0; i < 10; ++i) { for (int t = 0; t < 16; ++t) acc[i][t] = 0.f; fr[i] = *reinterpret_cast<const h8*>(ring + lane * 16u + i * 1024); }
    for (int e = 0; e < 8; ++e) xb[e] = (_Float16)(0.001f * (lane + e));
    uint4 sa = make_uint4(1, 2, 3, 4), sb = sa; uint2 sc = make_uint2(5, 6);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint8_t* cp = stream + (size_t)(it & 63) * 10240;
        const uint8_t* sl = ring + (it % 3) * 10240 + lane * 16u;
        uint8_t* sw = ring + ((it + 1) % 3) * 10240;
        if constexpr (MASK & 8) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        else { __builtin_amdgcn_sched_barrier(0); }
        if constexpr (MASK & 2) { *reinterpret_cast<uint4*>(sw + o0) = sa; *reinterpret_cast<uint4*>(sw + o1) = sb; *reinterpret_cast<uint2*>(sw + o2) = sc; }
        if constexpr (MASK & 4) { sa = *reinterpret_cast<const uint4*>(cp + o0); sb = *reinterpret_cast<const uint4*>(cp + o1); sc = *reinterpret_cast<const uint2*>(cp + o2); }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            if constexpr (!(MASK & 16)) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xb, acc[i], 0, 0, 0);
            if constexpr (MASK & 1) fr[i] = *reinterpret_cast<const h8*>(sl + i * 1024);
        }
        if constexpr (MASK & 2) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
        if constexpr (MASK & 4) __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
        if constexpr (!(MASK & 16)) {
#pragma unroll
            for (int i = 0; i < 10; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); if constexpr (MASK & 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 10; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
    s += (float)(sa.x + sb.y + sc.x) + (float)fr[3][2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The step with the REAL data flow: three staging register sets, global loads three steps ahead of the LDS write that consumes them, fragment reads one step ahead of the MFMAs.
// TURNS: the four waves write their share of a chunk in turns (behind MFMAs 1, 3, 5, 7) instead of all at the top of the step.
template <bool TURNS, bool WRITES, bool LOADS, bool READS, bool XB = false>
__global__ __launch_bounds__(256) void step3_kernel(const uint8_t* __restrict__ stream, float* out, uint64_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t r0[10240], r1[10240], r2[10240], padlds[60000];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t o0 = wave * 1024u + lane * 16u, o1 = o0 + 4096u, o2 = (8u + (wave >> 1)) * 1024u + (wave & 1u) * 512u + lane * 8u;
    for (uint32_t i = threadIdx.x; i < 10240 / 4; i += 256) { reinterpret_cast<uint32_t*>(r0)[i] = 0x3C003C00u; reinterpret_cast<uint32_t*>(r1)[i] = 0x3C003C00u; reinterpret_cast<uint32_t*>(r2)[i] = 0x3C003C00u; }
    if (iters < 0) padlds[threadIdx.x] = 1;
    __syncthreads();
    f16v acc[10];
    h8 fr[10], xb, xf[20];
    for (int i = 0; i < 10; ++i) { for (int t = 0; t < 16; ++t) acc[i][t] = 0.f; fr[i] = *reinterpret_cast<const h8*>(r0 + lane * 16u + i * 1024); }
    for (int e = 0; e < 8; ++e) xb[e] = (_Float16)(0.001f * (lane + e));
    for (int q = 0; q < 20; ++q) xf[q] = *reinterpret_cast<const h8*>(stream + lane * 16u + q * 1024);
    f16v S[3]; for (int q = 0; q < 3; ++q) for (int t = 0; t < 16; ++t) S[q][t] = 0.f;
    uint4 sa0 = make_uint4(1, 2, 3, 4), sa1 = sa0, sa2 = sa0, sb0 = sa0, sb1 = sa0, sb2 = sa0; uint2 sc0 = make_uint2(5, 6), sc1 = sc0, sc2 = sc0;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](auto K, int it) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;      // it % 3
        const uint8_t* cp = stream + (size_t)((it + 5) & 63) * 10240;
        uint8_t* sw = k == 0 ? r2 : k == 1 ? r0 : r1;            // slot of chunk it + 2
        const uint8_t* sl = (k == 0 ? r1 : k == 1 ? r2 : r0) + lane * 16u;      // slot of chunk it + 1
        __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0);
        auto wr = [&]() __attribute__((always_inline)) {
            *reinterpret_cast<uint4*>(sw + o0) = k == 0 ? sa2 : k == 1 ? sa0 : sa1; *reinterpret_cast<uint4*>(sw + o1) = k == 0 ? sb2 : k == 1 ? sb0 : sb1;
            *reinterpret_cast<uint2*>(sw + o2) = k == 0 ? sc2 : k == 1 ? sc0 : sc1;
        };
        auto mm = [&](int i) __attribute__((always_inline)) {
            if constexpr (XB) { if (k == 0) S[i % 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xf[(i + 7 * k) % 20], S[i % 3], 0, 0, 0);
                                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xf[(i + 7 * k) % 20], acc[i], 0, 0, 0); }
            else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xb, acc[i], 0, 0, 0);
            if constexpr (READS) fr[i] = *reinterpret_cast<const h8*>(sl + i * 1024);
        };
        if constexpr (WRITES && !TURNS) wr();
        mm(0); mm(1);
        if constexpr (WRITES && TURNS) { if (wave == 0u) wr(); }
        mm(2); mm(3);
        if constexpr (WRITES && TURNS) { if (wave == 1u) wr(); }
        mm(4); mm(5);
        if constexpr (WRITES && TURNS) { if (wave == 2u) wr(); }
        mm(6); mm(7);
        if constexpr (WRITES && TURNS) { if (wave == 3u) wr(); }
        if constexpr (LOADS) {
            const uint4 a_ = *reinterpret_cast<const uint4*>(cp + o0), b_ = *reinterpret_cast<const uint4*>(cp + o1); const uint2 c_ = *reinterpret_cast<const uint2*>(cp + o2);
            if constexpr (k == 0) { sa2 = a_; sb2 = b_; sc2 = c_; } else if constexpr (k == 1) { sa0 = a_; sb0 = b_; sc0 = c_; } else { sa1 = a_; sb1 = b_; sc1 = c_; }
        }
        mm(8); mm(9);
    };
    for (int it = 0; it < iters; it += 3) {
        step(std::integral_constant<int, 0>{}, it); step(std::integral_constant<int, 1>{}, it + 1); step(std::integral_constant<int, 2>{}, it + 2);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 10; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
    for (int q = 0; q < 3; ++q) for (int t = 0; t < 16; ++t) s += S[q][t];
    for (int q = 0; q < 20; ++q) s += (float)xf[q][1];
    s += (float)(sa0.x + sb1.y + sc2.x + sa1.z + sa2.w + sb0.x + sb2.y + sc0.y + sc1.x) + (float)fr[3][2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The step with the stream going global -> LDS by DMA (global_load_lds, 16 B per lane = one 1 KiB fragment per wave instruction; no staging registers, no ds_write):
// six slots, chunk j + 5 is requested in step j (its slot held chunk j - 1), the fragments of chunk j + 1 are read in step j behind a counted vmcnt wait + barrier.
// Fragment reads are inline asm (hipcc waits vmcnt(0) in front of every LDS read it can see while a DMA may be in flight).  Per wave and chunk: two whole fragments
// + half a fragment (lane halves of two waves share fragments 8 / 9), so every wave issues the same three instructions and one counted wait serves all.
template <bool DMA, int SPREAD>
__global__ __launch_bounds__(256) void step_dma_kernel(const uint8_t* __restrict__ stream, float* out, uint64_t* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[6 * 10240];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    for (uint32_t i = threadIdx.x; i < 6 * 10240 / 4; i += 256) reinterpret_cast<uint32_t*>(ring)[i] = 0x3C003C00u;
    __syncthreads();
    f16v acc[10];
    h8 fr[10], xb;
    for (int i = 0; i < 10; ++i) { for (int t = 0; t < 16; ++t) acc[i][t] = 0.f; fr[i] = *reinterpret_cast<const h8*>(ring + lane * 16u + i * 1024); }
    for (int e = 0; e < 8; ++e) xb[e] = (_Float16)(0.001f * (lane + e));
    const bool mine = (lane >> 5) == (wave & 1u);                       // this lane's half of the shared fragment
    const uint32_t g0 = wave * 1024u + lane * 16u, g1 = g0 + 4096u, g2 = (8u + (wave >> 1)) * 1024u + lane * 16u;
    const uint32_t l0 = wave * 1024u, l1 = l0 + 4096u, l2 = (8u + (wave >> 1)) * 1024u;
    uint32_t rd = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)ring + lane * 16u;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    auto step = [&](auto K, int it) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;      // it % 6
        const uint8_t* cp = stream + (size_t)((it + 5) & 63) * 10240;
        uint8_t* sw = ring + ((k + 5) % 6) * 10240;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DMA) asm volatile("s_waitcnt vmcnt(9)\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        auto dma = [&](int q) __attribute__((always_inline)) {
            if constexpr (DMA) {
                if (q == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cp + g0), (__attribute__((address_space(3))) void*)(sw + l0), 16, 0, 0);
                if (q == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cp + g1), (__attribute__((address_space(3))) void*)(sw + l1), 16, 0, 0);
                if (q == 2) { if (mine) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cp + g2), (__attribute__((address_space(3))) void*)(sw + l2), 16, 0, 0); }
            }
        };
        uint32_t rda = rd;
#define DMM(i) do { acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[i], xb, acc[i], 0, 0, 0); \
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[i]) : "v"(rda), "n"(((k + 1) % 6) * 10240 + (i) * 1024) : "memory"); } while (0)
        if constexpr (SPREAD == 0) { dma(0); dma(1); dma(2); }
        DMM(0); if constexpr (SPREAD == 1) dma(0);
        DMM(1); DMM(2); if constexpr (SPREAD == 1) dma(1);
        DMM(3); DMM(4); if constexpr (SPREAD == 1) dma(2);
        DMM(5); DMM(6); DMM(7); DMM(8); DMM(9);
#undef DMM
    };
    for (int it = 0; it < iters; it += 6) {
        step(std::integral_constant<int, 0>{}, it); step(std::integral_constant<int, 1>{}, it + 1); step(std::integral_constant<int, 2>{}, it + 2);
        step(std::integral_constant<int, 3>{}, it + 3); step(std::integral_constant<int, 4>{}, it + 4); step(std::integral_constant<int, 5>{}, it + 5);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 10; ++i) for (int t = 0; t < 16; ++t) s += acc[i][t];
    s += (float)fr[3][2];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool DMA, int SPREAD> void run_dma(const char* what, const uint8_t* stream, float* out, uint64_t* cyc) {
    const int iters = 3996;
    hipLaunchKernelGGL((step_dma_kernel<DMA, SPREAD>), dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((step_dma_kernel<DMA, SPREAD>), dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    printf("%-70s %7.1f cycles per step (s_memtime), %7.1f ns per step\n", what, mean / iters, ms * 1e6 / iters);
}

template <bool TURNS, bool WRITES, bool LOADS, bool READS, bool XB = false> void run3(const char* what, const uint8_t* stream, float* out, uint64_t* cyc) {
    const int iters = 3999;
    hipLaunchKernelGGL((step3_kernel<TURNS, WRITES, LOADS, READS, XB>), dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((step3_kernel<TURNS, WRITES, LOADS, READS, XB>), dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    printf("%-70s %7.1f cycles per step (s_memtime), %7.1f ns per step\n", what, mean / iters, ms * 1e6 / iters);
}

template <int MASK> void run(const char* what, const uint8_t* stream, float* out, uint64_t* cyc) {
    const int iters = 4000;
    hipLaunchKernelGGL(step_kernel<MASK>, dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(step_kernel<MASK>, dim3(256), dim3(256), 0, 0, stream, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= 256;
    printf("%-58s %7.1f cycles per step (s_memtime), %7.1f ns per step\n", what, mean / iters, ms * 1e6 / iters);
}

int main() {
    uint8_t* stream; float* out; uint64_t* cyc;
    hipMalloc(&stream, 64 * 10240 + 65536); hipMemset(stream, 0x3c, 64 * 10240 + 65536);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    run<0>("10 MFMA 32x32x16", stream, out, cyc);
    run<1>("10 MFMA + 10 fragment reads", stream, out, cyc);
    run<3>("10 MFMA + 10 fragment reads + 3 LDS writes", stream, out, cyc);
    run<7>("10 MFMA + reads + writes + 3 global loads", stream, out, cyc);
    run<15>("10 MFMA + reads + writes + loads + barrier", stream, out, cyc);
    run<9>("10 MFMA + reads + barrier", stream, out, cyc);
    run<8>("10 MFMA + barrier", stream, out, cyc);
    run<31>("no MFMA: reads + writes + loads + barrier", stream, out, cyc);
    run<17>("no MFMA: 10 fragment reads", stream, out, cyc);
    run<18>("no MFMA: 3 LDS writes", stream, out, cyc);
    run<20>("no MFMA: 3 global loads", stream, out, cyc);
    run<24>("no MFMA: barrier", stream, out, cyc);
    printf("-- the real data flow: loads three steps ahead, writes one step ahead, barrier per step --\n");
    run3<false, false, false, true>("10 MFMA + reads + barrier", stream, out, cyc);
    run3<false, true, false, true>("  + writes (all four waves at the top of the step)", stream, out, cyc);
    run3<true, true, false, true>("  + writes (the waves take turns)", stream, out, cyc);
    run3<false, false, true, true>("  + global loads (three steps of lead), no writes", stream, out, cyc);
    run3<false, true, true, true>("  + writes at the top + loads", stream, out, cyc);
    run3<true, true, true, true>("  + writes in turns + loads", stream, out, cyc);
    run3<false, true, true, true, true>("  writes at the top + loads, 20 B fragments + 3-accumulator chains", stream, out, cyc);
    run3<true, true, true, true, true>("  writes in turns + loads, 20 B fragments + 3-accumulator chains", stream, out, cyc);
    printf("-- the stream by LDS-DMA (global_load_lds b128): no staging registers, no ds_write; asm fragment reads, six slots, five steps of lead --\n");
    run_dma<false, 0>("10 MFMA + asm reads + barrier (no stream)", stream, out, cyc);
    run_dma<true, 0>("  + 3 DMA pieces per wave at the top of the step", stream, out, cyc);
    run_dma<true, 1>("  + 3 DMA pieces per wave behind MFMAs 0, 2, 4", stream, out, cyc);
    return 0;
}
