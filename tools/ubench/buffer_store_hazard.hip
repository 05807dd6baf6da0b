// A write-after-read window behind buffer_store_dwordx4 on gfx950: the wide tile's first straight-line epilogue stored 16 bytes per lane through a
// buffer descriptor and overwrote the data registers with the next block's VALU results a few instructions later; lanes 12-15 / 44-47 of the second dword arrived in
// memory with the NEW values.  This is the same sequence in isolation: per wave 20 stores of 16 bytes per lane, each followed at once by packed fp32 adds into the registers
// just stored; the host checks every stored dword against the value the registers held when the store was issued.
//   mode 0: buffer_store_dwordx4 with an SGPR soffset   1: buffer_store_dwordx4, soffset = 0 (the case LLVM's hazard recognizer pads)   2: global_store_dwordx4
//   gap: s_nop wait states between the store and the first overwriting VALU instruction
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/buffer_store_hazard.hip -o tools/ubench/bin/buffer_store_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, int GAP>
__global__ __launch_bounds__(512) void k(uint32_t* __restrict__ y, int tiles) {
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(y, 0, -1, 0x00020000);
    f4 v = {1.0f + lane, 2.0f + lane, 3.0f + lane, 4.0f + lane};
    const f4 inc = {0.5f, 0.25f, 0.125f, 1.0f};
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint32_t base = (uint32_t)(((size_t)t * 8 + wave) * 20 * 64 * 16);           // bytes: [tile][wave][20 stores][64 lanes][16 B]
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            u4 d = __builtin_bit_cast(u4, v);
            const uint32_t so = base + (uint32_t)i * 1024u;
            if (MODE == 0) __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)(lane * 16u), (int)so, 0);
            if (MODE == 1) __builtin_amdgcn_raw_buffer_store_b128(d, rs, (int)(lane * 16u + so), 0, 0);
            if (MODE == 2) *reinterpret_cast<u4*>(reinterpret_cast<uint8_t*>(y) + (size_t)so + lane * 16u) = d;
            if (GAP == 1) asm volatile("s_nop 0"); if (GAP == 4) asm volatile("s_nop 3"); if (GAP == 8) asm volatile("s_nop 7"); if (GAP == 16) asm volatile("s_nop 7\n s_nop 7");
            // overwrite the registers just stored (the compiler keeps v in place: d is a bit cast of it)
            asm volatile("v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3" : "+v"(*(reinterpret_cast<double*>(&v))), "+v"(*(reinterpret_cast<double*>(&v) + 1)) : "v"(*(reinterpret_cast<const double*>(&inc))), "v"(*(reinterpret_cast<const double*>(&inc) + 1)));
        }
    }
}
template <int MODE, int GAP> static void run(const char* name, uint32_t* y, int tiles) {
    const size_t n = (size_t)tiles * 8 * 20 * 64 * 4;
    hipMemset(y, 0xFF, n * 4);
    hipLaunchKernelGGL((k<MODE, GAP>), dim3(256), dim3(512), 0, 0, y, tiles);
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), y, n * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; int first = -1; size_t per_dword[4] = {0, 0, 0, 0}, per_lane16[16] = {0};
    for (int t = 0; t < tiles; ++t) {
        const int it0 = (t - (t % 256)) / 256;          // iteration of the workgroup that ran tile t: its registers advanced 20 steps per earlier tile
        for (int w = 0; w < 8; ++w) for (int i = 0; i < 20; ++i) for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) {
            const float inc[4] = {0.5f, 0.25f, 0.125f, 1.0f};
            float e = 1.0f + c + l;
            for (int s = 0; s < it0 * 20 + i; ++s) e += inc[c];
            const uint32_t got = h[((((size_t)t * 8 + w) * 20 + i) * 64 + l) * 4 + c];
            uint32_t eb; memcpy(&eb, &e, 4);
            if (got != eb) { ++bad; ++per_dword[c]; ++per_lane16[l & 15]; if (first < 0) first = l; }
        }
    }
    printf("%-46s gap %2d: %8zu wrong dwords of %zu", name, GAP, bad, n);
    if (bad) printf("  (by dword %zu %zu %zu %zu; lanes mod 16 with errors:", per_dword[0], per_dword[1], per_dword[2], per_dword[3]);
    if (bad) { for (int l = 0; l < 16; ++l) if (per_lane16[l]) printf(" %d", l); printf(")"); }
    printf("\n");
}
int main() {
    const int tiles = 512;
    uint32_t* y; hipMalloc(&y, (size_t)tiles * 8 * 20 * 64 * 16);
    run<0, 0>("buffer_store_dwordx4, SGPR soffset", y, tiles); run<0, 1>("buffer_store_dwordx4, SGPR soffset", y, tiles); run<0, 4>("buffer_store_dwordx4, SGPR soffset", y, tiles);
    run<0, 8>("buffer_store_dwordx4, SGPR soffset", y, tiles); run<0, 16>("buffer_store_dwordx4, SGPR soffset", y, tiles);
    run<1, 0>("buffer_store_dwordx4, soffset 0", y, tiles); run<1, 4>("buffer_store_dwordx4, soffset 0", y, tiles);
    run<2, 0>("global_store_dwordx4", y, tiles);
    return 0;
}
