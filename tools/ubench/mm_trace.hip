// Debug build of the matmul engine with cycle stamps (MM_TRACE): where does a step's time go?  Prints, for the waves of workgroup 0,
// per step: even-phase issue block | vmcnt wait | barrier | odd-phase issue block | barrier  (shader cycles).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mm_trace.hip -o tools/ubench/bin/mm_trace
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
    const long M = argc > 1 ? atol(argv[1]) : 32768; const int K = argc > 2 ? atoi(argv[2]) : 1280, N = argc > 3 ? atoi(argv[3]) : 2560;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;          // GSW_GEMM_PLAIN 0 / GEGLU 1 / TRANS 2 / TOK2PF 3
    const int use_bias = argc > 5 ? atoi(argv[5]) : 0;
    _Float16 *x, *w, *y; hipMalloc(&x, M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&y, (size_t)(M * 1.08) * N * 2 + (1 << 20));
    fill<<<1024, 256>>>(x, M * K, 1); fill<<<1024, 256>>>(w, (size_t)N * K, 2);
    _Float16* bias = nullptr; if (use_bias) { hipMalloc(&bias, (size_t)N * 2); fill<<<64, 256>>>(bias, N, 3); }
    unsigned long long* tb; hipMalloc(&tb, 4096); hipMemset(tb, 0, 4096);
    hipMemcpyToSymbol(HIP_SYMBOL(g_mm_trace_buf), &tb, sizeof(tb));
    for (int r = 0; r < 3; ++r) { int rc = gsw_gemm(x, w, bias, nullptr, y, M, K, N, mode, mode >= 2 ? 4096 : 0, mode == 3 ? 64 : 0, GSW_F16, nullptr); if (rc) { printf("rc %d\n", rc); return 1; } }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(12 * 16);
    hipMemcpy(h.data(), tb, h.size() * 8, hipMemcpyDeviceToHost);
    const int P = K / 64;
    const long tiles = ((M + 255) / 256) * (N / 160);
    const long my_tiles = (tiles + 255) / 256;            // workgroup 0
    const double steps = (double)my_tiles * P * 3;        // 3 launches accumulate
    printf("M=%ld K=%d N=%d: %d steps per tile, %ld tiles for workgroup 0; cycles per step: gap | even issue | vmcnt wait | lgkm+barrier | odd issue | lgkm+barrier ; epilogue per tile\n", M, K, N, P, my_tiles);
    for (int wv : {0, 3, 4, 7}) {
        const unsigned long long* t = &h[wv * 16];
        double tot = 0; for (int k = 0; k < 6; ++k) tot += t[k] / steps;
        printf("wave %d: %6.0f | %6.0f | %6.0f | %6.0f | %6.0f | %6.0f  = %6.0f per step ; epilogue %7.0f per tile\n", wv, t[0] / steps, t[1] / steps, t[2] / steps, t[3] / steps,
               t[4] / steps, t[5] / steps, tot, t[6] / (my_tiles * 3.0));
        const double nt = my_tiles * 3.0;
        printf("        epilogue per tile: tile decode %5.0f | row set-up %5.0f | column blocks 0 .. 4: %5.0f | %5.0f | %5.0f | %5.0f | %5.0f ; the rest %5.0f\n",
               t[13] / nt, t[14] / nt, t[8] / nt, t[9] / nt, t[10] / nt, t[11] / nt, t[12] / nt, (t[6] - 0.0) / nt);
    }
    for (int wv : {8, 11}) {                                  // SPLIT only: the producer waves' step = issue | wait for the previous stage to land | barrier
        const unsigned long long* t = &h[wv * 16];
        if (t[0]) printf("producer wave %d: issue %6.0f | vmcnt wait %6.0f | barrier wait %6.0f per stage\n", wv, t[0] / steps, t[1] / steps, t[2] / steps);
    }
    return 0;
}
