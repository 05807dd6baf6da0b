// gswm_image.hip -- image-side stages either side of the latent loops, gfx950.
//
//   X1   extract.py:31-43  load_image: PIL Lanczos resize -> ToTensor -> fp16 -> 2x-1         gsw_lanczos_plan + gsw_resize_lanczos
//   G1   modified_stable_diffusion_gs `decode_image` / `torch_to_numpy` / numpy_to_pil        gsw_tensor_to_image
//   D    distortions:131-233: JPEG quality QF ("compression"), Lanczos "scaling", brightness,
//        contrast, togray, invert, flips, additive Gaussian noise                            gsw_jpeg_roundtrip, gsw_image_pointwise
//
// All of it is byte / integer work bounded by HBM: one pass reads 3 B/pixel and writes 3 B/pixel (6 B when the output is the
// fp16 CHW tensor the VAE encoder consumes).  The arithmetic restates, bit for bit, what Pillow (Resample.c, Blend.c, Convert.c)
// and libjpeg(-turbo) (jccolor.c, jcsample.c, jfdctint.c, jcdctmgr.c, jidctint.c, jdsample.c, jdcolor.c) compute for the calls the
// reference makes; oracle/image_oracle.py is the CPU restatement the tests compare against (itself pinned to PIL).
//
// Kernels
//   gsw_resample_h_kernel / _v_kernel : Pillow's two-pass fixed-point resampler (22 fractional bits, uint8 between the passes);
//                                       one workgroup per image row, the source row staged once in LDS (horizontal pass),
//                                       coalesced column-parallel accumulation (vertical pass) with the output conversion fused.
//   gsw_jpeg_blocks_kernel            : one wave per 16x256 RGB tile staged in LDS (coalesced 16-byte loads); one thread per 8x8 block runs
//                                       the whole RGB->YCbCr(+2x2 box) -> FDCT -> quantise -> dequantise -> IDCT chain in 64 registers.
//   gsw_jpeg_finish_kernel            : triangle ("fancy") chroma upsampling + YCbCr->RGB + output conversion, one thread per pixel.
//   gsw_image_pointwise_kernel        : the point-wise attacks; contrast's mean grey level comes from gsw_image_lsum_kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>

#include "../../include/gswm.h"

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip; read by gsw_last_hip_error()
#define g_img_hip_error g_last_hip_error

#define GSW_IMG_LAUNCH_CHECK() do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) { g_img_hip_error = (int)_e; return GSW_ERR_HIP; } } while (0)

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;      // Resample.c

// ---------------------------------------------------------------------------------------------------------------------------
// output conversion shared by the resize / JPEG / point-wise kernels
// ---------------------------------------------------------------------------------------------------------------------------
// GSW_IMG_U8_HWC   : uint8 [B, H, W, 3]                                  (a PIL image)
// GSW_IMG_F16_CHW  : fp16  [B, 3, H, W] = fp16(2 * fp16(v / 255) - 1)    (extract.py:37,48,40: ToTensor -> .to(float16) -> 2.*x - 1.)
// GSW_IMG_F32_CHW  : fp32  [B, 3, H, W] = v / 255                        (ToTensor)
__device__ __forceinline__ void store_px(void* out, int mode, int64_t b, int y, int x, int H, int W, int c, int nch, uint32_t v) {
    if (mode == GSW_IMG_U8_HWC) {
        ((uint8_t*)out)[((b * H + y) * (int64_t)W + x) * nch + c] = (uint8_t)v;
    } else if (mode == GSW_IMG_F16_CHW) {
        const float f = (float)v / 255.0f;                        // ToTensor: float32 division
        const _Float16 h = (_Float16)f;                           // .to(dtype=float16)
        const _Float16 r = (_Float16)((float)((_Float16)(2.0f * (float)h)) - 1.0f);   // 2.*x (exact) then - 1. rounded to fp16
        ((_Float16*)out)[((b * nch + c) * (int64_t)H + y) * W + x] = r;
    } else {
        ((float*)out)[((b * nch + c) * (int64_t)H + y) * W + x] = (float)v / 255.0f;
    }
}

__device__ __forceinline__ int clip8(int32_t v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// ---------------------------------------------------------------------------------------------------------------------------
// Pillow resampler
// ---------------------------------------------------------------------------------------------------------------------------
// horizontal pass: in [rows, Win, nch] u8 -> out [rows, Wout, nch] u8; one workgroup per R consecutive rows (staged once in LDS), so
// that a thread's coefficient row k[0..xmax) is fetched once and applied to R pixels
constexpr int RESAMPLE_ROWS = 8;
__global__ __launch_bounds__(256) void gsw_resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                             const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                             int Win, int Wout, int nch, int64_t rows, int R) {
    extern __shared__ uint8_t row[];
    const int64_t r0 = (int64_t)blockIdx.x * R;
    const int nr = (int)min((int64_t)R, rows - r0);
    const int nb = Win * nch;                       // bytes per row; LDS pitch rounded up to 4
    const int pitch = (nb + 3) & ~3;
    const uint8_t* src = in + r0 * (int64_t)nb;
    if (((nb & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 3) == 0)) {
        const int nw = nb >> 2;
        for (int i = threadIdx.x; i < nw * nr; i += blockDim.x) {
            const int rr = i / nw, w = i - rr * nw;
            reinterpret_cast<uint32_t*>(row + rr * pitch)[w] = reinterpret_cast<const uint32_t*>(src + (int64_t)rr * nb)[w];
        }
    } else {
        for (int i = threadIdx.x; i < nb * nr; i += blockDim.x) {
            const int rr = i / nb, w = i - rr * nb;
            row[rr * pitch + w] = src[(int64_t)rr * nb + w];
        }
    }
    __syncthreads();
    uint8_t* dst = out + r0 * (int64_t)Wout * nch;
    for (int o = threadIdx.x; o < Wout * nch; o += blockDim.x) {
        const int xx = o / nch, c = o - xx * nch;
        const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
        const int32_t* k = kk + (int64_t)xx * ksize;
        int32_t ss[RESAMPLE_ROWS];
#pragma unroll
        for (int rr = 0; rr < RESAMPLE_ROWS; ++rr) ss[rr] = 1 << (PRECISION_BITS - 1);
        const uint8_t* p = row + xmin * nch + c;
        for (int x = 0; x < xmax; ++x) {
            const int32_t kv = k[x];
#pragma unroll
            for (int rr = 0; rr < RESAMPLE_ROWS; ++rr)
                if (rr < R) ss[rr] += (int32_t)p[rr * pitch + x * nch] * kv;        // rows in [nr, R) read stale LDS, never stored
        }
#pragma unroll
        for (int rr = 0; rr < RESAMPLE_ROWS; ++rr)
            if (rr < nr) dst[(int64_t)rr * Wout * nch + o] = (uint8_t)clip8(ss[rr] >> PRECISION_BITS);
    }
}

// vertical pass: in [B, Hin, W, nch] u8 -> out (mode) [B, Hout, W]; grid (Hout, B)
__global__ __launch_bounds__(256) void gsw_resample_v_kernel(const uint8_t* __restrict__ in, void* __restrict__ out,
                                                             const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                             int Hin, int Hout, int W, int nch, int mode) {
    const int yy = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
    const int32_t* k = kk + (int64_t)yy * ksize;
    const int pitch = W * nch;
    const uint8_t* src = in + (b * Hin + ymin) * (int64_t)pitch;
    for (int o = threadIdx.x; o < pitch; o += blockDim.x) {
        int32_t ss = 1 << (PRECISION_BITS - 1);
        for (int y = 0; y < ymax; ++y) ss += (int32_t)src[(int64_t)y * pitch + o] * k[y];
        const int x = o / nch, c = o - x * nch;
        store_px(out, mode, b, yy, x, Hout, W, c, nch, (uint32_t)clip8(ss >> PRECISION_BITS));
    }
}

// no vertical resampling needed: plain conversion of a u8 HWC image
__global__ __launch_bounds__(256) void gsw_image_convert_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, int H, int W, int nch, int mode) {
    const int y = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int pitch = W * nch;
    const uint8_t* src = in + (b * H + y) * (int64_t)pitch;
    for (int o = threadIdx.x; o < pitch; o += blockDim.x) {
        const int x = o / nch, c = o - x * nch;
        store_px(out, mode, b, y, x, H, W, c, nch, src[o]);
    }
}

// fp16 / fp32 / bf16 [B, 3, H, W] in [0, 1]  ->  u8 [B, H, W, 3]: numpy_to_pil's (x * 255).round().astype(uint8) on the fp32 value
template <typename T>
__global__ __launch_bounds__(256) void gsw_tensor_to_image_kernel(const T* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int nch,
                                                                  int denorm) {
    const int y = blockIdx.x;
    const int64_t b = blockIdx.y;
    for (int o = threadIdx.x; o < W * nch; o += blockDim.x) {
        const int x = o / nch, c = o - x * nch;
        T t = in[((b * nch + c) * (int64_t)H + y) * W + x];
        if (denorm) {                        // decode_image: (image / 2 + 0.5).clamp(0, 1), evaluated in the tensor's dtype
            t = (T)((float)t * 0.5f);
            t = (T)((float)t + 0.5f);
            t = (T)fminf(fmaxf((float)t, 0.0f), 1.0f);
        }
        const float v = rintf((float)t * 255.0f);                  // fp32 multiply, round half to even
        out[((b * H + y) * (int64_t)W + x) * nch + c] = (uint8_t)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// JPEG lossy stages
// ---------------------------------------------------------------------------------------------------------------------------
struct JpegTables {
    uint8_t q[2][64];      // luma, chroma quantisation tables in natural (row-major) order
    uint32_t magic[2][64]; // ceil(2^32 / (8 q)): n / (8 q) == umulhi(n, magic) exactly for n < 2^21 (8 q <= 2040)
};

constexpr int CONST_BITS = 13, PASS1_BITS = 2;
constexpr int32_t FIX_0_298631336 = 2446, FIX_0_390180644 = 3196, FIX_0_541196100 = 4433, FIX_0_765366865 = 6270, FIX_0_899976223 = 7373,
                  FIX_1_175875602 = 9633, FIX_1_501321110 = 12299, FIX_1_847759065 = 15137, FIX_1_961570560 = 16069, FIX_2_053119869 = 16819,
                  FIX_2_562915447 = 20995, FIX_3_072711026 = 25172;

__device__ __forceinline__ int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c: one 8-point pass; FIRST = row pass (scale up by PASS1_BITS), else column pass (remove it, keep the factor 8)
template <bool FIRST>
__device__ __forceinline__ void fdct8(int32_t& d0, int32_t& d1, int32_t& d2, int32_t& d3, int32_t& d4, int32_t& d5, int32_t& d6, int32_t& d7) {
    int32_t tmp0 = d0 + d7, tmp7 = d0 - d7, tmp1 = d1 + d6, tmp6 = d1 - d6, tmp2 = d2 + d5, tmp5 = d2 - d5, tmp3 = d3 + d4, tmp4 = d3 - d4;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    constexpr int SH = FIRST ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS;
    if (FIRST) { d0 = (tmp10 + tmp11) << PASS1_BITS; d4 = (tmp10 - tmp11) << PASS1_BITS; }
    else { d0 = descale(tmp10 + tmp11, PASS1_BITS); d4 = descale(tmp10 - tmp11, PASS1_BITS); }
    int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
    d2 = descale(z1 + tmp13 * FIX_0_765366865, SH);
    d6 = descale(z1 + tmp12 * (-FIX_1_847759065), SH);
    z1 = tmp4 + tmp7;
    int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    const int32_t z5 = (z3 + z4) * FIX_1_175875602;
    tmp4 *= FIX_0_298631336; tmp5 *= FIX_2_053119869; tmp6 *= FIX_3_072711026; tmp7 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 = z3 * (-FIX_1_961570560) + z5;
    z4 = z4 * (-FIX_0_390180644) + z5;
    d7 = descale(tmp4 + z1 + z3, SH);
    d5 = descale(tmp5 + z2 + z4, SH);
    d3 = descale(tmp6 + z2 + z3, SH);
    d1 = descale(tmp7 + z1 + z4, SH);
}

// jidctint.c: one 8-point pass; FIRST = column pass, else row pass (descale by CONST_BITS + PASS1_BITS + 3)
template <bool FIRST>
__device__ __forceinline__ void idct8(int32_t& i0, int32_t& i1, int32_t& i2, int32_t& i3, int32_t& i4, int32_t& i5, int32_t& i6, int32_t& i7) {
    int32_t z2 = i2, z3 = i6;
    int32_t z1 = (z2 + z3) * FIX_0_541196100;
    const int32_t tmp2 = z1 + z3 * (-FIX_1_847759065), tmp3 = z1 + z2 * FIX_0_765366865;
    const int32_t tmp0 = (i0 + i4) << CONST_BITS, tmp1 = (i0 - i4) << CONST_BITS;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int32_t t0 = i7, t1 = i5, t2 = i3, t3 = i1;
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
    int32_t z4 = t1 + t3;
    const int32_t z5 = (z3 + z4) * FIX_1_175875602;
    t0 *= FIX_0_298631336; t1 *= FIX_2_053119869; t2 *= FIX_3_072711026; t3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 = z3 * (-FIX_1_961570560) + z5;
    z4 = z4 * (-FIX_0_390180644) + z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    constexpr int SH = FIRST ? CONST_BITS - PASS1_BITS : CONST_BITS + PASS1_BITS + 3;
    i0 = descale(tmp10 + t3, SH); i7 = descale(tmp10 - t3, SH);
    i1 = descale(tmp11 + t2, SH); i6 = descale(tmp11 - t2, SH);
    i2 = descale(tmp12 + t1, SH); i5 = descale(tmp12 - t1, SH);
    i3 = descale(tmp13 + t0, SH); i4 = descale(tmp13 - t0, SH);
}

#define ROW8(F, d, r) F(d[(r) * 8 + 0], d[(r) * 8 + 1], d[(r) * 8 + 2], d[(r) * 8 + 3], d[(r) * 8 + 4], d[(r) * 8 + 5], d[(r) * 8 + 6], d[(r) * 8 + 7])
#define COL8(F, d, c) F(d[0 * 8 + (c)], d[1 * 8 + (c)], d[2 * 8 + (c)], d[3 * 8 + (c)], d[4 * 8 + (c)], d[5 * 8 + (c)], d[6 * 8 + (c)], d[7 * 8 + (c)])

// FDCT -> quantise -> dequantise -> IDCT of one block held in registers; d: level-shifted samples in, reconstructed samples (0..255) out
__device__ __forceinline__ void jpeg_block_codec(int32_t (&d)[64], const uint8_t* __restrict__ q, const uint32_t* __restrict__ magic) {
#pragma unroll
    for (int r = 0; r < 8; ++r) ROW8(fdct8<true>, d, r);
#pragma unroll
    for (int c = 0; c < 8; ++c) COL8(fdct8<false>, d, c);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        // jcdctmgr.c quantize: round-half-away of coef / (8 q); then the decoder's dequantisation (coef * q)
        const int32_t qq = (int32_t)q[i];
        const int32_t qv = qq << 3;
        const int32_t a = (d[i] < 0 ? -d[i] : d[i]) + (qv >> 1);
        const int32_t m = (int32_t)__umulhi((uint32_t)a, magic[i]);          // == a / qv (a < 2^18), without the ~40-instruction integer divide
        d[i] = (d[i] < 0 ? -m : m) * qq;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) COL8(idct8<true>, d, c);
#pragma unroll
    for (int r = 0; r < 8; ++r) ROW8(idct8<false>, d, r);
#pragma unroll
    for (int i = 0; i < 64; ++i) d[i] = clip8(d[i] + 128);
}

// jccolor.c rgb_ycc_convert
__device__ __forceinline__ int32_t ycc_y(int32_t r, int32_t g, int32_t b) { return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16; }
__device__ __forceinline__ int32_t ycc_cb(int32_t r, int32_t g, int32_t b) { return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16; }
__device__ __forceinline__ int32_t ycc_cr(int32_t r, int32_t g, int32_t b) { return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16; }

// One wave per 16-row x 256-column tile of the image (one MCU row x 16 MCUs): the RGB tile is staged in LDS with coalesced 16-byte
// loads (byte loads with clamping for tiles that touch the right edge of an image whose width is not a multiple of 16, or an
// unaligned base), replicated at the image edges as libjpeg pads (last column / last row).  Then one thread per 8x8 block runs the
// whole chain in registers: all 64 lanes a luma block, lanes 0-31 then a Cb / Cr block.  grid = (tiles_x, tiles_y, B).
// Planes: yplane [B][nby*8][nbx*8], cplane [B][2][ncy*8][ncx*8].
constexpr int JT_W = 256, JT_H = 16, JT_PITCH = JT_W * 3 + 16;      // +16: rows start 16-byte aligned, bank-shifted by 4 words
__global__ __launch_bounds__(64) void gsw_jpeg_blocks_kernel(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ yplane, uint8_t* __restrict__ cplane,
                                                             JpegTables tb, int H, int W, int nby, int nbx, int ncy, int ncx) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[JT_H * JT_PITCH];
    const int64_t b = blockIdx.z;
    const int tx = blockIdx.x, ty = blockIdx.y, lane = threadIdx.x;
    const int x0 = tx * JT_W, y0 = ty * JT_H;
    const uint8_t* img = rgb + b * (int64_t)H * W * 3;
    const bool fast = (x0 + JT_W <= W) && ((W & 15) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    if (fast) {
        for (int i = lane; i < JT_H * (JT_W * 3 / 16); i += 64) {                // 16 rows x 48 chunks
            const int r = i / 48, c = i - r * 48;
            const int y = min(y0 + r, H - 1);                                     // bottom edge: replicate the last row
            *reinterpret_cast<uint4*>(tile + r * JT_PITCH + c * 16) = *reinterpret_cast<const uint4*>(img + ((int64_t)y * W + x0) * 3 + c * 16);
        }
    } else {
        for (int i = lane; i < JT_H * JT_W; i += 64) {
            const int r = i / JT_W, c = i - r * JT_W;
            const int y = min(y0 + r, H - 1), x = min(x0 + c, W - 1);             // right edge: replicate the last column
            const uint8_t* q = img + ((int64_t)y * W + x) * 3;
            uint8_t* o = tile + r * JT_PITCH + c * 3;
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
        }
    }
    __syncthreads();
    int32_t d[64];
    {   // luma: lane -> block (2 ty + lane / 32, 32 tx + lane % 32)
        const int byl = lane >> 5, bxl = lane & 31;
        const int by = ty * 2 + byl, bx = tx * 32 + bxl;
        if (by < nby && bx < nbx) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const uint8_t* p = tile + (byl * 8 + r) * JT_PITCH + bxl * 24;
#pragma unroll
                for (int c = 0; c < 8; ++c) d[r * 8 + c] = ycc_y(p[c * 3], p[c * 3 + 1], p[c * 3 + 2]) - 128;
            }
            jpeg_block_codec(d, tb.q[0], tb.magic[0]);
            uint8_t* o = yplane + (b * nby * 8 + by * 8) * (int64_t)(nbx * 8) + bx * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint32_t lo = d[r * 8] | (d[r * 8 + 1] << 8) | (d[r * 8 + 2] << 16) | (d[r * 8 + 3] << 24);
                uint32_t hi = d[r * 8 + 4] | (d[r * 8 + 5] << 8) | (d[r * 8 + 6] << 16) | (d[r * 8 + 7] << 24);
                *(uint2*)(o + (int64_t)r * nbx * 8) = make_uint2(lo, hi);
            }
        }
    }
    if (lane < 32) {   // chroma: lane -> component lane / 16, block (ty, 16 tx + lane % 16)
        const int comp = lane >> 4, bxl = lane & 15;
        const int by = ty, bx = tx * 16 + bxl;
        const int hc = (H + 1) >> 1;
        if (by < ncy && bx < ncx) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // rows past the downsampled image replicate ITS last row (jcprepct.c expand_bottom_edge on the output buffer);
                // the full-resolution rows are only padded to a pair (one row group) -- the tile already replicates row H-1
                const int j = min(by * 8 + r, hc - 1);
                const uint8_t* p0 = tile + (2 * j - y0) * JT_PITCH + bxl * 48;
                const uint8_t* p1 = p0 + JT_PITCH;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const uint8_t* a0 = p0 + c * 6;                               // pixels 2i, 2i+1 of both rows (right edge replicated in the tile)
                    const uint8_t* a1 = p1 + c * 6;
                    int32_t sum;
                    if (comp == 0)
                        sum = ycc_cb(a0[0], a0[1], a0[2]) + ycc_cb(a0[3], a0[4], a0[5]) + ycc_cb(a1[0], a1[1], a1[2]) + ycc_cb(a1[3], a1[4], a1[5]);
                    else
                        sum = ycc_cr(a0[0], a0[1], a0[2]) + ycc_cr(a0[3], a0[4], a0[5]) + ycc_cr(a1[0], a1[1], a1[2]) + ycc_cr(a1[3], a1[4], a1[5]);
                    d[r * 8 + c] = ((sum + 1 + (c & 1)) >> 2) - 128;              // h2v2_downsample: bias 1, 2, 1, 2 ...
                }
            }
            jpeg_block_codec(d, tb.q[1], tb.magic[1]);
            uint8_t* o = cplane + ((b * 2 + comp) * ncy * 8 + by * 8) * (int64_t)(ncx * 8) + bx * 8;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                uint32_t lo = d[r * 8] | (d[r * 8 + 1] << 8) | (d[r * 8 + 2] << 16) | (d[r * 8 + 3] << 24);
                uint32_t hi = d[r * 8 + 4] | (d[r * 8 + 5] << 8) | (d[r * 8 + 6] << 16) | (d[r * 8 + 7] << 24);
                *(uint2*)(o + (int64_t)r * ncx * 8) = make_uint2(lo, hi);
            }
        }
    }
}

// jdsample.c h2v2_fancy_upsample (or plain replication when the downsampled width is <= 2) + jdcolor.c ycc_rgb_convert
__global__ __launch_bounds__(256) void gsw_jpeg_finish_kernel(const uint8_t* __restrict__ yplane, const uint8_t* __restrict__ cplane, void* __restrict__ out,
                                                              int H, int W, int nby, int nbx, int ncy, int ncx, int mode) {
    const int y = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int hc = (H + 1) >> 1, wc = (W + 1) >> 1;
    const int j = y >> 1, v = y & 1;
    const int jn = v == 0 ? max(j - 1, 0) : min(j + 1, hc - 1);       // the further row; beyond the image = the edge row itself
    const int pc = ncx * 8;
    const uint8_t* yrow = yplane + (b * nby * 8 + y) * (int64_t)(nbx * 8);
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        const int i = x >> 1, h = x & 1;
        int32_t cc[2];
#pragma unroll
        for (int comp = 0; comp < 2; ++comp) {
            const uint8_t* cp = cplane + ((b * 2 + comp) * ncy * 8) * (int64_t)pc;
            if (wc > 2) {
                const int32_t cur = 3 * (int32_t)cp[(int64_t)j * pc + i] + (int32_t)cp[(int64_t)jn * pc + i];
                if (h == 0) {
                    if (i == 0) cc[comp] = (cur * 4 + 8) >> 4;
                    else cc[comp] = (cur * 3 + 3 * (int32_t)cp[(int64_t)j * pc + i - 1] + (int32_t)cp[(int64_t)jn * pc + i - 1] + 8) >> 4;
                } else {
                    if (i == wc - 1) cc[comp] = (cur * 4 + 7) >> 4;
                    else cc[comp] = (cur * 3 + 3 * (int32_t)cp[(int64_t)j * pc + i + 1] + (int32_t)cp[(int64_t)jn * pc + i + 1] + 7) >> 4;
                }
            } else {
                cc[comp] = cp[(int64_t)j * pc + i];
            }
        }
        const int32_t yy = yrow[x], cb = cc[0] - 128, cr = cc[1] - 128;
        const int32_t r = clip8(yy + ((91881 * cr + 32768) >> 16));
        const int32_t g = clip8(yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
        const int32_t bl = clip8(yy + ((116130 * cb + 32768) >> 16));
        store_px(out, mode, b, y, x, H, W, 0, 3, (uint32_t)r);
        store_px(out, mode, b, y, x, H, W, 1, 3, (uint32_t)g);
        store_px(out, mode, b, y, x, H, W, 2, 3, (uint32_t)bl);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// point-wise attacks
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rgb2l(uint32_t r, uint32_t g, uint32_t b) { return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16; }   // Convert.c

// per-image sum of L over all pixels (exact, uint64): ImageStat.Stat(image.convert("L")).mean[0] = sum / count.
// 16 pixels (48 bytes = three 16-byte loads) per thread and iteration; one atomic per workgroup.
__global__ __launch_bounds__(256) void gsw_image_lsum_kernel(const uint8_t* __restrict__ in, unsigned long long* __restrict__ sums, int64_t npix) {
    __shared__ unsigned long long part[4];
    const int64_t b = blockIdx.y;
    const uint8_t* img = in + b * npix * 3;
    unsigned long long s = 0;
    const int64_t ngrp = ((reinterpret_cast<uintptr_t>(img) & 15) == 0) ? npix / 16 : 0;      // aligned groups of 16 pixels
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < ngrp; g += (int64_t)gridDim.x * blockDim.x) {
        const uint4* q = reinterpret_cast<const uint4*>(img + g * 48);
        const uint4 a = q[0], c = q[1], d = q[2];
        const uint32_t w[12] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
        uint32_t acc = 0;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
            const int o = px * 3;
            const uint32_t r = (w[o >> 2] >> ((o & 3) * 8)) & 255u, gg = (w[(o + 1) >> 2] >> (((o + 1) & 3) * 8)) & 255u,
                           bl = (w[(o + 2) >> 2] >> (((o + 2) & 3) * 8)) & 255u;
            acc += rgb2l(r, gg, bl);
        }
        s += acc;
    }
    for (int64_t p = ngrp * 16 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x)
        s += rgb2l(img[p * 3], img[p * 3 + 1], img[p * 3 + 2]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(&sums[b], t);
    }
}

// Blend.c ImagingBlend(in1 = degenerate, in2 = image, alpha): float arithmetic, truncation; clipped outside [0, 1]
__device__ __forceinline__ uint32_t blend_u8(int32_t a, int32_t v, float alpha) {
#pragma clang fp contract(off)                                                     // C evaluates the multiply, then the add: no FMA
    const float prod = alpha * (float)(v - a);
    const float t = (float)a + prod;
    if (alpha >= 0.0f && alpha <= 1.0f) return (uint32_t)(uint8_t)(int32_t)t;
    return t <= 0.0f ? 0u : (t >= 255.0f ? 255u : (uint32_t)(int32_t)t);
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// op: GSW_PW_*; one thread per pixel
__global__ __launch_bounds__(256) void gsw_image_pointwise_kernel(const uint8_t* __restrict__ in, void* __restrict__ out, int H, int W, int op, float strength,
                                                                  const unsigned long long* __restrict__ lsums, uint64_t seed, uint64_t image_index0,
                                                                  int mode) {
    const int y = blockIdx.x;
    const int64_t b = blockIdx.y;
    const uint8_t* img = in + b * (int64_t)H * W * 3;
    int32_t grey = 0;
    if (op == GSW_PW_CONTRAST) {
        const double mean = (double)lsums[b] / (double)((int64_t)H * W);
        grey = (int32_t)(mean + 0.5);                                   // ImageEnhance.Contrast: int(mean + 0.5)
    }
    for (int x = threadIdx.x; x < W; x += blockDim.x) {
        int sy = y, sx = x;
        if (op == GSW_PW_HFLIP) sx = W - 1 - x;
        if (op == GSW_PW_VFLIP) sy = H - 1 - y;
        const uint8_t* p = img + ((int64_t)sy * W + sx) * 3;
        uint32_t r = p[0], g = p[1], bl = p[2];
        if (op == GSW_PW_BRIGHTNESS) { r = blend_u8(0, r, strength); g = blend_u8(0, g, strength); bl = blend_u8(0, bl, strength); }
        else if (op == GSW_PW_CONTRAST) { r = blend_u8(grey, r, strength); g = blend_u8(grey, g, strength); bl = blend_u8(grey, bl, strength); }
        else if (op == GSW_PW_INVERT) { r = 255u - r; g = 255u - g; bl = 255u - bl; }
        else if (op == GSW_PW_GRAY) { r = g = bl = rgb2l(r, g, bl); }
        else if (op == GSW_PW_NOISE) {
            // v/255 + std * N(0,1), clamp to [0,1], back to uint8 by rounding (distortions:166-173); Box-Muller on Philox words keyed
            // by (seed; pixel index, global image index): independent of batch split and GPU count
            uint32_t w[4];
            const uint64_t pix = (uint64_t)y * W + x, gi = image_index0 + (uint64_t)b;
            philox4x32_10((uint32_t)pix, (uint32_t)(pix >> 32), (uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
            const float u0 = ((float)(w[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(w[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((float)(w[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(w[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
            const float n0 = r0 * cosf(6.2831853071795864f * u1), n1 = r0 * sinf(6.2831853071795864f * u1), n2 = r1 * cosf(6.2831853071795864f * u3);
            const float fr = fminf(fmaxf((float)r / 255.0f + strength * n0, 0.0f), 1.0f);
            const float fg = fminf(fmaxf((float)g / 255.0f + strength * n1, 0.0f), 1.0f);
            const float fb = fminf(fmaxf((float)bl / 255.0f + strength * n2, 0.0f), 1.0f);
            r = (uint32_t)rintf(fr * 255.0f); g = (uint32_t)rintf(fg * 255.0f); bl = (uint32_t)rintf(fb * 255.0f);
        }
        store_px(out, mode, b, y, x, H, W, 0, 3, r);
        store_px(out, mode, b, y, x, H, W, 1, 3, g);
        store_px(out, mode, b, y, x, H, W, 2, 3, bl);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Gaussian blur: PIL's ImageFilter.GaussianBlur = three passes per axis of an "extended box" filter (BoxBlur.c): out = (ww * sum of the
// 2r+1 window + fw * (the two pixels just outside it) + 2^23) >> 24 in uint32 arithmetic, edges replicated, uint8 between passes.
// Pillow slides an accumulator along each line; the sum is exact integer arithmetic, so evaluating the window per output pixel gives
// the same bytes and parallelises over every pixel.
// ---------------------------------------------------------------------------------------------------------------------------
// horizontal pass: one workgroup per row (staged in LDS), thread per output byte
__global__ __launch_bounds__(256) void gsw_boxblur_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int W, int radius, uint32_t ww, uint32_t fw) {
    extern __shared__ uint8_t row[];
    const int64_t r = blockIdx.x;
    const int nb = W * 3;
    const uint8_t* src = in + r * nb;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) row[i] = src[i];
    __syncthreads();
    uint8_t* dst = out + r * nb;
    for (int o = threadIdx.x; o < nb; o += blockDim.x) {
        const int x = o / 3, c = o - x * 3;
        uint32_t acc = 0;
        for (int d = -radius; d <= radius; ++d) acc += row[min(max(x + d, 0), W - 1) * 3 + c];
        const uint32_t far = (uint32_t)row[min(max(x - radius - 1, 0), W - 1) * 3 + c] + (uint32_t)row[min(max(x + radius + 1, 0), W - 1) * 3 + c];
        dst[o] = (uint8_t)((acc * ww + far * fw + (1u << 23)) >> 24);
    }
}

// vertical pass: grid (H, B), thread per output byte of the row; window rows are read from global (coalesced along the row, L2-resident)
__global__ __launch_bounds__(256) void gsw_boxblur_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int radius, uint32_t ww,
                                                            uint32_t fw) {
    const int y = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int nb = W * 3;
    const uint8_t* img = in + b * (int64_t)H * nb;
    uint8_t* dst = out + (b * H + y) * (int64_t)nb;
    for (int o = threadIdx.x; o < nb; o += blockDim.x) {
        uint32_t acc = 0;
        for (int d = -radius; d <= radius; ++d) acc += img[(int64_t)min(max(y + d, 0), H - 1) * nb + o];
        const uint32_t far = (uint32_t)img[(int64_t)min(max(y - radius - 1, 0), H - 1) * nb + o] + (uint32_t)img[(int64_t)min(max(y + radius + 1, 0), H - 1) * nb + o];
        dst[o] = (uint8_t)((acc * ww + far * fw + (1u << 23)) >> 24);
    }
}

double sinc_filter(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return std::sin(x) / x;
}

double lanczos_filter(double x) {
    if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
    return 0.0;
}

bool mode_ok(int m) { return m == GSW_IMG_U8_HWC || m == GSW_IMG_F16_CHW || m == GSW_IMG_F32_CHW; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// host C ABI
// ---------------------------------------------------------------------------------------------------------------------------
int gsw_lanczos_plan(int in_size, int out_size, int32_t* bounds, int32_t* kk, int kk_capacity) {
    // Resample.c precompute_coeffs (double, libm sin) + normalize_coeffs_8bpc for the whole-image box.  Returns ksize (> 0), or a
    // negative gsw_status.  bounds == kk == NULL: size query only.
    if (in_size <= 0 || out_size <= 0) return -GSW_ERR_BAD_ARG;
    double scale, filterscale;
    filterscale = scale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    if (!bounds && !kk) return ksize;
    if (!bounds || !kk || (int64_t)kk_capacity < (int64_t)out_size * ksize) return -GSW_ERR_BAD_ARG;
    const double ss = 1.0 / filterscale;
    std::unique_ptr<double[]> w(new double[ksize]);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        int32_t* k = kk + (int64_t)xx * ksize;
        for (int x = 0; x < xmax; ++x) {
            w[x] = lanczos_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) w[x] /= ww;
            k[x] = w[x] < 0 ? (int32_t)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int32_t)(0.5 + w[x] * (1 << PRECISION_BITS));
        }
        for (int x = xmax; x < ksize; ++x) k[x] = 0;
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return ksize;
}

int gsw_resize_lanczos(const uint8_t* in_dev, int B, int Hin, int Win, void* out_dev, int Hout, int Wout, int out_mode, uint8_t* tmp_dev,
                       const int32_t* hbounds_dev, const int32_t* hkk_dev, int hksize, const int32_t* vbounds_dev, const int32_t* vkk_dev, int vksize,
                       void* stream) {
    if (!in_dev || !out_dev || B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || !mode_ok(out_mode)) return GSW_ERR_BAD_ARG;
    const bool need_h = Wout != Win, need_v = Hout != Hin;                       // Resample.c ImagingResample: a pass is skipped when the size is kept
    if ((need_h && (!hbounds_dev || !hkk_dev || hksize <= 0)) || (need_v && (!vbounds_dev || !vkk_dev || vksize <= 0))) return GSW_ERR_BAD_ARG;
    if ((need_h && (need_v || out_mode != GSW_IMG_U8_HWC)) && !tmp_dev) return GSW_ERR_BAD_ARG;
    if ((int64_t)Win * 3 > 160 * 1024 || (int64_t)B * Hin > 0x7FFFFFFF) return GSW_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const uint8_t* src = in_dev;
    if (need_h) {
        // straight into out_dev when nothing follows; else into tmp_dev [B, Hin, Wout, 3]
        uint8_t* dst = (!need_v && out_mode == GSW_IMG_U8_HWC) ? (uint8_t*)out_dev : tmp_dev;
        const int R = (int)std::max<int64_t>(1, std::min<int64_t>(RESAMPLE_ROWS, (48 * 1024) / (((int64_t)Win * 3 + 3) & ~3)));
        const size_t lds = (size_t)R * (((size_t)Win * 3 + 3) & ~(size_t)3);
        const int64_t rows = (int64_t)B * Hin;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)gsw_resample_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { g_img_hip_error = (int)e; return GSW_ERR_HIP; }
        }
        hipLaunchKernelGGL(gsw_resample_h_kernel, dim3((uint32_t)((rows + R - 1) / R)), dim3(256), lds, st, in_dev, dst, hbounds_dev, hkk_dev, hksize, Win, Wout, 3, rows, R);
        GSW_IMG_LAUNCH_CHECK();
        if (dst == (uint8_t*)out_dev) return GSW_OK;
        src = tmp_dev;
    }
    if (need_v)
        hipLaunchKernelGGL(gsw_resample_v_kernel, dim3(Hout, B), dim3(256), 0, st, src, out_dev, vbounds_dev, vkk_dev, vksize, Hin, Hout, Wout, 3, out_mode);
    else
        hipLaunchKernelGGL(gsw_image_convert_kernel, dim3(Hout, B), dim3(256), 0, st, src, out_dev, Hout, Wout, 3, out_mode);
    GSW_IMG_LAUNCH_CHECK();
    return GSW_OK;
}

int gsw_tensor_to_image(const void* in_dev, int dtype, int B, int H, int W, int denormalise, uint8_t* out_dev, void* stream) {
    if (!in_dev || !out_dev || B <= 0 || H <= 0 || W <= 0) return GSW_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GSW_F16) hipLaunchKernelGGL(gsw_tensor_to_image_kernel<_Float16>, dim3(H, B), dim3(256), 0, st, (const _Float16*)in_dev, out_dev, H, W, 3, denormalise);
    else if (dtype == GSW_BF16) hipLaunchKernelGGL(gsw_tensor_to_image_kernel<__bf16>, dim3(H, B), dim3(256), 0, st, (const __bf16*)in_dev, out_dev, H, W, 3, denormalise);
    else if (dtype == GSW_F32) hipLaunchKernelGGL(gsw_tensor_to_image_kernel<float>, dim3(H, B), dim3(256), 0, st, (const float*)in_dev, out_dev, H, W, 3, denormalise);
    else return GSW_ERR_BAD_ARG;
    GSW_IMG_LAUNCH_CHECK();
    return GSW_OK;
}

int gsw_jpeg_quant_tables(int quality, uint8_t luma[64], uint8_t chroma[64]) {
    // jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline = TRUE), natural order
    static const uint8_t std_l[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                      18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
    static const uint8_t std_c[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
    if (!luma || !chroma) return GSW_ERR_BAD_ARG;
    int q = quality <= 0 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = q < 50 ? 5000 / q : 200 - q * 2;
    for (int i = 0; i < 64; ++i) {
        long l = ((long)std_l[i] * scale + 50L) / 100L, c = ((long)std_c[i] * scale + 50L) / 100L;
        luma[i] = (uint8_t)(l <= 0 ? 1 : (l > 255 ? 255 : l));
        chroma[i] = (uint8_t)(c <= 0 ? 1 : (c > 255 ? 255 : c));
    }
    return GSW_OK;
}

size_t gsw_jpeg_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nby = (size_t)(H + 7) / 8, nbx = (size_t)(W + 7) / 8;
    const size_t ncy = (size_t)((H + 1) / 2 + 7) / 8, ncx = (size_t)((W + 1) / 2 + 7) / 8;
    return (size_t)B * (nby * nbx + 2 * ncy * ncx) * 64;
}

int gsw_jpeg_roundtrip(const uint8_t* rgb_dev, int B, int H, int W, int quality, void* out_dev, int out_mode, uint8_t* workspace_dev, void* stream) {
    if (!rgb_dev || !out_dev || !workspace_dev || B <= 0 || H <= 0 || W <= 0 || !mode_ok(out_mode)) return GSW_ERR_BAD_ARG;
    if (B > 65535 || H > 65500 || W > 65500) return GSW_ERR_UNSUPPORTED;         // JPEG's own dimension limit
    JpegTables tb;
    gsw_jpeg_quant_tables(quality, tb.q[0], tb.q[1]);
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 64; ++i) {
            const uint64_t dv = (uint64_t)tb.q[t][i] * 8u;
            tb.magic[t][i] = (uint32_t)((((uint64_t)1 << 32) + dv - 1) / dv);
        }
    const int nby = (H + 7) / 8, nbx = (W + 7) / 8, ncy = ((H + 1) / 2 + 7) / 8, ncx = ((W + 1) / 2 + 7) / 8;
    uint8_t* yplane = workspace_dev;
    uint8_t* cplane = workspace_dev + (size_t)B * nby * nbx * 64;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gsw_jpeg_blocks_kernel, dim3((W + JT_W - 1) / JT_W, (H + JT_H - 1) / JT_H, B), dim3(64), 0, st, rgb_dev, yplane, cplane, tb, H, W, nby, nbx,
                       ncy, ncx);
    GSW_IMG_LAUNCH_CHECK();
    hipLaunchKernelGGL(gsw_jpeg_finish_kernel, dim3(H, B), dim3(256), 0, st, (const uint8_t*)yplane, (const uint8_t*)cplane, out_dev, H, W, nby, nbx, ncy, ncx, out_mode);
    GSW_IMG_LAUNCH_CHECK();
    return GSW_OK;
}

int gsw_image_pointwise(const uint8_t* rgb_dev, int B, int H, int W, int op, float strength, uint64_t seed, uint64_t image_index0, void* out_dev,
                        int out_mode, uint64_t* workspace_dev, void* stream) {
    if (!rgb_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || !mode_ok(out_mode) || op < GSW_PW_BRIGHTNESS || op > GSW_PW_NOISE) return GSW_ERR_BAD_ARG;
    if (op == GSW_PW_CONTRAST && !workspace_dev) return GSW_ERR_BAD_ARG;        // [B] uint64
    if (B > 65535) return GSW_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (op == GSW_PW_CONTRAST) {
        hipError_t e = hipMemsetAsync(workspace_dev, 0, (size_t)B * sizeof(uint64_t), st);
        if (e != hipSuccess) { g_img_hip_error = (int)e; return GSW_ERR_HIP; }
        const int64_t npix = (int64_t)H * W;
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(64, (npix / 16 + 255) / 256));
        hipLaunchKernelGGL(gsw_image_lsum_kernel, dim3(nb, B), dim3(256), 0, st, rgb_dev, (unsigned long long*)workspace_dev, npix);
        GSW_IMG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(gsw_image_pointwise_kernel, dim3(H, B), dim3(256), 0, st, rgb_dev, out_dev, H, W, op, strength, (const unsigned long long*)workspace_dev,
                       seed, image_index0, out_mode);
    GSW_IMG_LAUNCH_CHECK();
    return GSW_OK;
}

int gsw_gaussian_blur_params(float radius, int passes, int* box_radius, uint32_t* ww, uint32_t* fw) {
    // BoxBlur.c _gaussian_blur_radius + the weights of ImagingHorizontalBoxBlur, with the C expression's float / double mix
    if (!box_radius || !ww || !fw || radius < 0 || passes < 1) return GSW_ERR_BAD_ARG;
    float sigma2, L, l, a;
    sigma2 = radius * radius / passes;
    L = sqrt(12.0 * sigma2 + 1.0);
    l = floor((L - 1.0) / 2.0);
    a = (2 * l + 1) * (l * (l + 1) - 3 * sigma2);
    a /= 6 * (sigma2 - (l + 1) * (l + 1));
    const float floatRadius = l + a;
    const int r = (int)floatRadius;
    const uint32_t w = (uint32_t)((uint32_t)(1 << 24) / (floatRadius * 2 + 1));
    *box_radius = r;
    *ww = w;
    *fw = ((1u << 24) - (uint32_t)(r * 2 + 1) * w) / 2;
    return GSW_OK;
}

int gsw_gaussian_blur(const uint8_t* rgb_dev, int B, int H, int W, float radius, uint8_t* out_dev, uint8_t* tmp_dev, void* stream) {
    // distortions:157-164 `image.filter(ImageFilter.GaussianBlur(radius))`: uint8 [B, H, W, 3] -> out_dev; tmp_dev: same size scratch
    if (!rgb_dev || !out_dev || !tmp_dev || B <= 0 || H <= 0 || W <= 0 || radius < 0) return GSW_ERR_BAD_ARG;
    if (B > 65535 || (int64_t)W * 3 > 160 * 1024 || (int64_t)B * H > 0x7FFFFFFF) return GSW_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)B * H * W * 3;
    if (radius == 0.0f) {                                   // ImageFilter.GaussianBlur(0): a copy
        hipError_t e = hipMemcpyAsync(out_dev, rgb_dev, bytes, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { g_img_hip_error = (int)e; return GSW_ERR_HIP; }
        return GSW_OK;
    }
    int r;
    uint32_t ww, fw;
    gsw_gaussian_blur_params(radius, 3, &r, &ww, &fw);
    const size_t lds = (size_t)W * 3;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gsw_boxblur_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { g_img_hip_error = (int)e; return GSW_ERR_HIP; }
    }
    const dim3 gh((uint32_t)((int64_t)B * H)), gv(H, B);
    // ImagingBoxBlur: three horizontal passes, then three vertical ones; ping-pong so that the last pass lands in out_dev
    hipLaunchKernelGGL(gsw_boxblur_h_kernel, gh, dim3(256), lds, st, rgb_dev, tmp_dev, W, r, ww, fw);
    hipLaunchKernelGGL(gsw_boxblur_h_kernel, gh, dim3(256), lds, st, (const uint8_t*)tmp_dev, out_dev, W, r, ww, fw);
    hipLaunchKernelGGL(gsw_boxblur_h_kernel, gh, dim3(256), lds, st, (const uint8_t*)out_dev, tmp_dev, W, r, ww, fw);
    hipLaunchKernelGGL(gsw_boxblur_v_kernel, gv, dim3(256), 0, st, (const uint8_t*)tmp_dev, out_dev, H, W, r, ww, fw);
    hipLaunchKernelGGL(gsw_boxblur_v_kernel, gv, dim3(256), 0, st, (const uint8_t*)out_dev, tmp_dev, H, W, r, ww, fw);
    hipLaunchKernelGGL(gsw_boxblur_v_kernel, gv, dim3(256), 0, st, (const uint8_t*)tmp_dev, out_dev, H, W, r, ww, fw);
    GSW_IMG_LAUNCH_CHECK();
    return GSW_OK;
}
