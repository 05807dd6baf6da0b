// Cycle stamps of the matmul engine's WIDE tile (256 x 320, MM_TRACE build): where does a tile's time go?  Per tile of workgroup 0, for waves 0, 1, 6, 7:
// step 0 up to its barrier | wait at that barrier | steady-state steps: issue, vmcnt wait, barrier wait | the tail in front of the epilogue | the epilogue.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mm_trace_wide.hip -o tools/ubench/bin/mm_trace_wide
//   usage: mm_trace_wide M K N mode(0 plain, 1 GEGLU) bias(0/1) resid(0/1)
#define MM_TRACE 1
__attribute__((visibility("hidden"))) thread_local int g_last_hip_error = 0;
#include "../../a-watermark-for-diffusion-models_amd/csrc/gswm_mm.hip"
#include <stdio.h>
#include <vector>
__global__ void fill(_Float16* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; p[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)); }
}
int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 524288; const int K = argc > 2 ? atoi(argv[2]) : 320, N = argc > 3 ? atoi(argv[3]) : 2560;
    const int mode = argc > 4 ? atoi(argv[4]) : 1, use_bias = argc > 5 ? atoi(argv[5]) : 1, use_res = argc > 6 ? atoi(argv[6]) : 0;
    _Float16 *x, *w, *y, *r = nullptr; hipMalloc(&x, M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&y, (size_t)M * N * 2 + (1 << 20));
    fill<<<1024, 256>>>(x, M * K, 1); fill<<<1024, 256>>>(w, (size_t)N * K, 2);
    _Float16* bias = nullptr; if (use_bias) { hipMalloc(&bias, (size_t)N * 2); fill<<<64, 256>>>(bias, N, 3); }
    if (use_res) { hipMalloc(&r, (size_t)M * N * 2); fill<<<1024, 256>>>(r, (size_t)M * N, 4); }
    unsigned long long* tb; hipMalloc(&tb, 4096); hipMemset(tb, 0, 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_mm_trace_buf), &tb, sizeof(tb));
    gsw_mm_config(512, -1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 3;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) { int rc = gsw_gemm(x, w, bias, r, y, M, K, N, mode, 0, 0, GSW_F16, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(12 * 16);
    hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
    const int P = K / 64;
    const long tiles = ((M + 255) / 256) * ((N + 319) / 320);
    const double nt = (double)((tiles + 255) / 256);            // tiles of workgroup 0 per launch (the last launch's sums are what the buffer holds)
    printf("M=%ld K=%d N=%d mode=%d bias=%d resid=%d: %.1f us per launch, %d steps per tile, %.0f tiles per workgroup\n", M, K, N, mode, use_bias, use_res, ms * 1e3 / reps, P, nt);
    printf("cycles per tile: step 0 issue | first barrier | steady steps (P-1): issue + vmcnt wait + barrier wait | last tail | epilogue: up to its vmcnt(0) + that wait + the rest (stores) || sum\n");
    for (int wv = 0; wv < 8; ++wv) {
        const unsigned long long* t = &h[wv * 16];
        double s = 0; for (int k = 0; k < 9; ++k) s += t[k] / nt;
        printf("wave %d: %7.0f | %6.0f | %7.0f + %6.0f + %6.0f | %6.0f | %6.0f + %6.0f + %6.0f || %8.0f per tile\n", wv, t[0] / nt, t[1] / nt, t[2] / nt, t[3] / nt, t[4] / nt, t[5] / nt, t[7] / nt, t[8] / nt, t[6] / nt, s);
    }
    return 0;
}
