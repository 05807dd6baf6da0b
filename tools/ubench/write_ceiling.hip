// Streaming-write ceiling of one MI355X for the embed kernel's store pattern: 1 GiB of fp32 written as 16-byte stores, no compute.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/write_ceiling tools/ubench/write_ceiling.hip && tools/ubench/bin/write_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>   // 0 plain, 1 nontemporal, 2 embed-like 2D grid (chunk of 2048 elems per WG per image, images strided over gridDim.y)
__global__ __launch_bounds__(256) void fill(float4* out, size_t n4, uint32_t N, int B) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (MODE == 2) {
        const uint32_t chunk = (blockIdx.x + blockIdx.y) % gridDim.x;
        for (int b = blockIdx.y; b < B; b += gridDim.y)
            for (int r = 0; r < 2; ++r) {
                const size_t off = ((size_t)b * N + chunk * 2048u + r * 1024u + 4u * threadIdx.x) >> 2;
                out[off] = v;
            }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            if (MODE == 1) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f4*>(out + i)); else out[i] = v;
        }
    }
}

// MODE 3: one workgroup owns whole images (64 KiB contiguous each), images strided over the grid; R = stores in flight per thread per burst
template <int R>
__global__ __launch_bounds__(256) void fill_img(float4* out, uint32_t N, int B) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const uint32_t n4 = N / 4;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        float4* o = out + (size_t)b * n4;
        for (uint32_t i = threadIdx.x; i < n4; i += 256 * R) {
#pragma unroll
            for (int r = 0; r < R; ++r) o[i + 256 * r] = v;
        }
    }
}

int main() {
    const int B = 16384; const uint32_t N = 16384;
    const size_t n4 = (size_t)B * N / 4;
    float4* d; hipMalloc(&d, n4 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %7.1f GB/s\n", name, ms / 20 * 1e3, n4 * 16.0 / (ms / 20 * 1e-3) / 1e9);
    };
    for (int g : {2048, 8192, 65536})
        run(g == 2048 ? "grid-stride plain, 2048 WGs" : g == 8192 ? "grid-stride plain, 8192 WGs" : "grid-stride plain, 65536 WGs", [&] { hipLaunchKernelGGL(fill<0>, dim3(g), dim3(256), 0, 0, d, n4, N, B); });
    for (int g : {2048, 8192, 65536})
        run(g == 2048 ? "grid-stride nontemporal, 2048 WGs" : g == 8192 ? "grid-stride nontemporal, 8192 WGs" : "grid-stride nontemporal, 65536 WGs", [&] { hipLaunchKernelGGL(fill<1>, dim3(g), dim3(256), 0, 0, d, n4, N, B); });
    run("embed-shaped grid (8 x 256), plain", [&] { hipLaunchKernelGGL(fill<2>, dim3(8, 256), dim3(256), 0, 0, d, n4, N, B); });
    run("embed-shaped grid (8 x 2048), plain", [&] { hipLaunchKernelGGL(fill<2>, dim3(8, 2048), dim3(256), 0, 0, d, n4, N, B); });
    for (int g : {2048, 4096, 16384}) {
        char nm[64]; snprintf(nm, 64, "image per WG (64 KiB contiguous), %d WGs, R=4", g);
        run(nm, [&] { hipLaunchKernelGGL(fill_img<4>, dim3(g), dim3(256), 0, 0, d, N, B); });
        snprintf(nm, 64, "image per WG (64 KiB contiguous), %d WGs, R=16", g);
        run(nm, [&] { hipLaunchKernelGGL(fill_img<16>, dim3(g), dim3(256), 0, 0, d, N, B); });
    }
    run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, n4 * 16, 0); });
    return 0;
}
