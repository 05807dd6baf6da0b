// Microbenchmark: the rate a bare MFMA stream sustains for SECONDS (register operands only, no memory), by MFMA shape and by operand content, with the shader clock
// and board power of THIS device sampled from sysfs meanwhile.  Question: is the matrix pipe's sustained rate on this board set by the clock the power
// management allows for the operand data (toggle rate), and does the 32x32x16 shape (half the A/B register reads per flop) sustain more than 16x16x32?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_power.hip -o tools/ubench/bin/mfma_power -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <glob.h>
#include <atomic>
#include <thread>
#include <vector>
#include <chrono>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32, 16 independent accumulators (4 A x 4 B fragments); 1: 32x32x16, 4 accumulators (2 A x 2 B)
__global__ __launch_bounds__(256) void k(const h8* __restrict__ src, float* out, int iters) {
    h8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(i * 256 + threadIdx.x)]; b[i] = src[((4 + i) * 256 + threadIdx.x)]; }
    float s = 0.f;
    if (SHAPE == 0) {
        f4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        f16v acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)          // two rounds over the four accumulators = the flops of the sixteen 16x16x32 above
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2 * r], b[j + 2 * r], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][7];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static std::atomic<bool> g_stop{false};
static std::vector<double> g_mhz, g_w;
static char g_clk[512], g_pw[512];
static void sampler() {
    while (!g_stop.load()) {
        FILE* f = fopen(g_clk, "r"); long v = 0;
        if (f) { if (fscanf(f, "%ld", &v) == 1) g_mhz.push_back(v / 1e6); fclose(f); }
        f = fopen(g_pw, "r");
        if (f) { if (fscanf(f, "%ld", &v) == 1) g_w.push_back(v / 1e6); fclose(f); }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
}

template <int SHAPE>
static void run(const char* name, const h8* src, float* out, int wps) {
    const int blocks = 256 * wps, iters = 40000;
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, out, 100);
    hipDeviceSynchronize();
    g_mhz.clear(); g_w.clear(); g_stop = false;
    std::thread th(sampler);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);                                  // calibrate: launches for ~3 s of steady load (the power management settles within ~0.5 s)
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float cal = 0; hipEventElapsedTime(&cal, e0, e1);
    const int reps = (int)(3000.0f / (cal / 4.0f)) + 1;
    g_mhz.clear(); g_w.clear();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    g_stop = true; th.join();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 4 /*waves*/ * iters * 16.0 * 16384.0;
    double mhz = 0, w = 0; size_t n0 = g_mhz.size() / 4;
    for (size_t i = n0; i < g_mhz.size(); ++i) mhz += g_mhz[i];
    for (size_t i = n0; i < g_w.size(); ++i) w += g_w[i];
    printf("%-52s %d wave/SIMD  %7.0f TFLOP/s  (%.2f s)  clock %5.0f MHz  power %5.0f W\n", name, wps, flops / ms / 1e9, ms / 1e3, g_mhz.size() > n0 ? mhz / (g_mhz.size() - n0) : 0.0,
           g_w.size() > n0 ? w / (g_w.size() - n0) : 0.0);
    fflush(stdout);
}

int main() {
    char bus[64]; hipDeviceGetPCIBusId(bus, sizeof bus, 0);
    for (char* c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c += 32;
    char pat[256]; glob_t g;
    snprintf(pat, sizeof pat, "/sys/bus/pci/devices/%s/hwmon/hwmon*/freq1_input", bus);
    g_clk[0] = g_pw[0] = 0;
    if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc) strncpy(g_clk, g.gl_pathv[0], sizeof g_clk - 1);
    snprintf(pat, sizeof pat, "/sys/bus/pci/devices/%s/hwmon/hwmon*/power1_input", bus);
    if (glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc) strncpy(g_pw, g.gl_pathv[0], sizeof g_pw - 1);
    snprintf(pat, sizeof pat, "/sys/bus/pci/devices/%s/hwmon/hwmon*/power1_average", bus);
    if (!g_pw[0] && glob(pat, 0, nullptr, &g) == 0 && g.gl_pathc) strncpy(g_pw, g.gl_pathv[0], sizeof g_pw - 1);
    printf("device %s  clock sensor %s  power sensor %s\n", bus, g_clk[0] ? g_clk : "(none)", g_pw[0] ? g_pw : "(none)");
    const size_t n = 8 * 256;
    std::vector<uint16_t> hr(n * 8), hz(n * 8, 0), hs(n * 8);
    srand(1);
    for (auto& v : hr) { float f = ((rand() & 0xffff) / 65536.f - 0.5f) * 4.f; _Float16 h = (_Float16)f; memcpy(&v, &h, 2); }       // random values in (-2, 2): every mantissa bit toggles
    for (auto& v : hs) { _Float16 h = (_Float16)1.0f; memcpy(&v, &h, 2); }                                                          // all ones: constant operands, non-zero products
    h8 *dr, *dz, *ds; float* out;
    hipMalloc(&dr, n * 16); hipMalloc(&dz, n * 16); hipMalloc(&ds, n * 16); hipMalloc(&out, 256 * 256 * 4 * sizeof(float));
    hipMemcpy(dr, hr.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(dz, hz.data(), n * 16, hipMemcpyHostToDevice); hipMemcpy(ds, hs.data(), n * 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 2; ++wps) {
        run<0>("16x16x32 f16, random operands", dr, out, wps);
        run<1>("32x32x16 f16, random operands", dr, out, wps);
        run<0>("16x16x32 f16, all-ones operands", ds, out, wps);
        run<1>("32x32x16 f16, all-ones operands", ds, out, wps);
        run<0>("16x16x32 f16, all-zero operands", dz, out, wps);
        run<1>("32x32x16 f16, all-zero operands", dz, out, wps);
    }
    return 0;
}
