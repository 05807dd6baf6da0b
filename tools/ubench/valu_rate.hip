// Issue rate of the VALU instructions the engine's epilogues are made of, per SIMD: one wave per SIMD (256 threads per workgroup, one workgroup per CU) runs a long
// unrolled stream of INDEPENDENT instructions of one kind; cycles per instruction = s_memtime delta / count.  (v_pk_fma_f32 does two FMAs per lane: the question is
// whether it issues in the 4 cycles of a v_fma_f32 or in 8.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_rate.hip -o tools/ubench/bin/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, c = {1.0001f, 0.9999f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (KIND == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));) }
        if (KIND == 2) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 3) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 4) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %0, %1\n v_cvt_pk_f16_f32 %1, %1, %2\n v_cvt_pk_f16_f32 %2, %2, %3\n v_cvt_pk_f16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 5) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));) }
        if (KIND == 6) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %4, %4\n v_pk_fma_f16 %1, %1, %4, %4\n v_pk_fma_f16 %2, %2, %4, %4\n v_pk_fma_f16 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p3[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 64.0 * iters;
    printf("%-18s %6.2f s_memtime ticks per instruction, %6.2f ns per instruction (one wave per SIMD)\n", name, c / n, ms * 1e6 / n);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<0>("v_fma_f32", out, cyc); run<1>("v_pk_fma_f32", out, cyc); run<5>("v_pk_mul_f32", out, cyc); run<6>("v_pk_fma_f16", out, cyc); run<2>("v_exp_f32", out, cyc); run<3>("v_rcp_f32", out, cyc); run<4>("v_cvt_pk_f16_f32", out, cyc);
    return 0;
}
