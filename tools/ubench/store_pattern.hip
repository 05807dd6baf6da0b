// How fast does a CU get an output tile out, as a function of the bytes each store instruction puts into one 128-byte line?  The matmul engine's dense-row epilogue
// (csrc/gswm_mm.hip, EPI 0) stores a wave's 128 x 80 fp16 block as 20 instructions of 64 lanes x 16 bytes; today an instruction covers 32 rows x 32 bytes
// (pattern 32).  Patterns: 32 = today's; 64 = 16 rows x 64 bytes (column blocks paired instead of row tiles; the fifth block stays at 32); 128 = 8 rows x 128 bytes
// (what a transpose through LDS could give; here only the address pattern, 96 of 80 columns clipped to the same byte count).  Same bytes, same grid (one 512-thread
// workgroup per CU, 256 x 320 tiles walked like the engine walks them), optional residual read with the same pattern.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/store_pattern.hip -o tools/ubench/bin/store_pattern ; usage: store_pattern [M] [N] [workgroups]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
template <int PAT, bool RES>
__global__ __launch_bounds__(512) void k(uint16_t* __restrict__ y, const uint16_t* __restrict__ r, int M, int N, int tiles_n, int ntiles) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wm = wave & 1u, grp = wave >> 1;
    const uint32_t q = lane >> 4, li = lane & 15u;
    uint4 v = make_uint4(lane, wave, 3, 4);
    for (uint32_t t = blockIdx.x; t < (uint32_t)ntiles; t += gridDim.x) {
        const uint32_t tm = t / tiles_n, tn = t - tm * tiles_n;
        const size_t row0 = (size_t)tm * 256 + wm * 128, col0 = (size_t)tn * 320 + grp * 80;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            size_t row, col;
            if (PAT == 32) { const int in = i % 5, pr = i / 5; row = (2 * pr + (q & 1u)) * 16 + li; col = in * 16 + (q >> 1) * 8; }
            else if (PAT == 64) {
                if (i < 16) { const int cp = i & 1, im = i >> 1; row = im * 16 + li; col = (cp * 2 + (q & 1u)) * 16 + (q >> 1) * 8; }
                else { const int pr = i - 16; row = (2 * pr + (q & 1u)) * 16 + li; col = 64 + (q >> 1) * 8; }
            } else if (PAT == 2) {       // lane PAIRS hold adjacent 16-byte chunks: 32 rows x 32 B per instruction, like 32 but with the chunk index in lane bit 0
                const int in = i % 5, pr = i / 5; row = (2 * pr + (q & 1u)) * 16 + (li & 14u) + (q >> 1); col = in * 16 + (li & 1u) * 8;
            } else if (PAT == 4) {       // lane QUADS hold four adjacent chunks: 16 rows x 64 B per instruction (blocks 0..3), the fifth block as pattern 2
                if (i < 16) { const int cp = i & 1, im = i >> 1; row = im * 16 + (li & 12u) + q; col = cp * 32 + (li & 3u) * 8; }
                else { const int pr = i - 16; row = (2 * pr + (q & 1u)) * 16 + (li & 14u) + (q >> 1); col = 64 + (li & 1u) * 8; }
            } else {      // 128: 8 rows x 128 B per instruction, 20 instructions = 160 rows x 64 columns' worth of bytes: rows 0..127 x 64 columns, then rows 0..31 again at column 64 (clipped to 16 columns x 4 passes)
                if (i < 16) { row = i * 8 + (lane >> 3); col = (lane & 7u) * 8; }
                else { row = (i - 16) * 32 + (lane >> 1); col = 64 + (lane & 1u) * 8; }
            }
            const size_t off = (row0 + row) * (size_t)N + col0 + col;
            if (RES) { const uint4 a = *reinterpret_cast<const uint4*>(r + off); v.x += a.x; v.y ^= a.y; v.z += a.z; v.w ^= a.w; }
            *reinterpret_cast<uint4*>(y + off) = v;
        }
    }
}
static int g_grid = 256;
template <int PAT, bool RES> static float run(uint16_t* y, const uint16_t* r, int M, int N, int reps) {
    const int tiles_n = N / 320, ntiles = (M / 256) * tiles_n;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<PAT, RES>), dim3(g_grid), dim3(512), 0, 0, y, r, M, N, tiles_n, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<PAT, RES>), dim3(g_grid), dim3(512), 0, 0, y, r, M, N, tiles_n, ntiles);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 131072, N = argc > 2 ? atoi(argv[2]) : 1280;
    g_grid = argc > 3 ? atoi(argv[3]) : 256;
    uint16_t *y, *r; const size_t bytes = (size_t)M * N * 2;
    hipMalloc(&y, bytes); hipMalloc(&r, bytes); hipMemset(r, 1, bytes);
    printf("output %d x %d fp16 (%.0f MB), %d workgroups x 512 threads, 256 x 320 tiles (%.1f per workgroup); per tile a CU stores 160 KiB\n", M, N, bytes / 1e6, g_grid, (M / 256) * (N / 320) / (double)g_grid);
    const float a = run<32, false>(y, r, M, N, 20), b = run<64, false>(y, r, M, N, 20), c = run<128, false>(y, r, M, N, 20);
    printf("store only      : 32 B per line and instruction %.1f us (%.2f TB/s) | 64 B %.1f us (%.2f) | 128 B %.1f us (%.2f)\n", a * 1e3, bytes / a / 1e9, b * 1e3, bytes / b / 1e9, c * 1e3, bytes / c / 1e9);
    { const float p2 = run<2, false>(y, r, M, N, 20), p4 = run<4, false>(y, r, M, N, 20), q2 = run<2, true>(y, r, M, N, 20), q4 = run<4, true>(y, r, M, N, 20);
      printf("adjacent lanes hold adjacent chunks: pairs (32 B) store %.1f us, with residual %.1f us | quads (64 B) store %.1f us, with residual %.1f us\n", p2 * 1e3, q2 * 1e3, p4 * 1e3, q4 * 1e3); }
    const float d = run<32, true>(y, r, M, N, 20), e = run<64, true>(y, r, M, N, 20), f = run<128, true>(y, r, M, N, 20);
    printf("residual + store: 32 B %.1f us (%.2f TB/s read + write) | 64 B %.1f us (%.2f) | 128 B %.1f us (%.2f)\n", d * 1e3, 2 * bytes / d / 1e9, e * 1e3, 2 * bytes / e / 1e9, f * 1e3, 2 * bytes / f / 1e9);
    return 0;
}
