// gswm_small.hip -- the small-batch regime of the eps model (rows X2 / G1 of SURVEY.md section 8a): the reference inverts ONE latent per call
// (extract.py:112-117) and BASELINE configs[1] is batch 8, where a UNet forward is ~500 dependent launches of 5-20 us, each paying the ~4.7 us
// floor of a dependent kernel boundary plus its own fill and drain.  Here live the kernels whose only purpose is FEWER, SHORTER launches at 1-16 rows:
//   * gsw_groupnorm_pf_fused : GroupNorm (+ SiLU) of a PF tensor (or of the channel concatenation of two) in ONE launch, one workgroup per (image,
//                              group): the group's values stay in registers between the statistics and the normalisation (exact two-pass variance);
//   * gsw_gather_rows        : out[b] = table[index[b]] -- the time-embedding chain of the UNet as a table over the training timesteps;
//   * gsw_nchw_to_pf         : latent [B, C, H, W] -> zero-bordered PF tensor with the channels zero-padded to one 64-wide K block;
//   * gsw_conv3x3_pf_nchw    : the 4-channel output convolution of the UNet (PF in, NCHW out), MFMA with the output channels zero-padded to 16.
// gfx950 only.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <algorithm>

#include "../../include/gswm.h"

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip
#define GSW_SM_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { g_last_hip_error = (int)e_; return GSW_ERR_HIP; } } while (0)

namespace {

__device__ __forceinline__ float sm_up(uint16_t h, bool bf) { return bf ? __uint_as_float((uint32_t)h << 16) : __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t sm_cvt(float f, bool bf) {
    if (bf) {                                          // round to nearest even (NaN stays NaN), like __float2bfloat16
        const uint32_t u = __float_as_uint(f);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40u);
        return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    return __half_as_ushort(__float2half_rn(f));
}

// sum over the workgroup in a FIXED order (lanes by xor butterflies, waves in index order): every thread returns the same value; s_red: >= 16 floats
__device__ __forceinline__ float sm_block_sum(float v, float* s_red, int nwaves) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                                   // s_red may still be read by a previous call
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = s_red[0];
    for (int w = 1; w < nwaves; ++w) t += s_red[w];
    return t;
}

// ------------------------------------------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU), one workgroup per (group, image).  A UNIT is one interior pixel x one channel PAIR of the group (4 bytes).  Thread layout:
// (pair pr, column xl, row lane rl) = (t % npair, (t / npair) % W, t / (npair W)); the thread owns pixels (rl + k RL, xl), k < H / RL <= MAXU, of its pair --
// a constant address stride per unit, no division -- held in registers: pass 1 the mean, pass 2 the centred variance (exact two-pass, no
// E[x^2] - mean^2 cancellation), pass 3 normalise + affine + SiLU + store.  The zero border of a PF output is written by the same workgroup for
// its channel slice.  Sources: x (Ca channels, or all C when x2 == null) | x2 (C - Ca channels).
// ------------------------------------------------------------------------------------------------------------------------------------------------
template <int MAXU>
__global__ __launch_bounds__(1024) void gsw_gn_fused_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int32_t Ca, const uint16_t* __restrict__ gamma,
                                                           const uint16_t* __restrict__ beta, uint16_t* __restrict__ y, int32_t C, int32_t G, int32_t H, int32_t W,
                                                           int32_t npair, int32_t RL, float eps, int act, int tokens, int bf) {
    __shared__ float s_red[16];
    const int32_t g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int32_t cpg = C / G, Hp = H + 2, Wp = W + 2, HW = H * W, HpWp = Hp * Wp;
    const int32_t pr = tid % npair, pl = tid / npair;
    const int32_t xl = pl % W, rl = pl / W;
    const int32_t nwaves = ((int32_t)blockDim.x + 63) >> 6;
    const int32_t c = g * cpg + 2 * pr;                               // first channel of this thread's pair
    const bool mine = rl < RL;                                        // (threads past npair * W * RL idle: blockDim is a multiple of 64)
    const int32_t rl0 = mine ? rl : 0;
    // Loads are UNCONDITIONAL (units past the image re-read the first one; their values are masked out of the sums and never stored) and
    // addressed as (wave-uniform base of unit k) + (32-bit lane offset): branch-free code with scalar per-unit bases keeps the 64 values in
    // registers -- one branch, or one 64-bit vector address, per unit made hipcc spill.  A group that straddles the two sources loads from both.
    const bool g_second = x2 != nullptr && g * cpg >= Ca;             // (wave-uniform) the whole group lives in x2
    const bool g_straddle = x2 != nullptr && g * cpg < Ca && (g + 1) * cpg > Ca;
    const int32_t ld1 = x2 ? Ca : C, ld2 = C - Ca;
    uint32_t v[MAXU];
    auto load_all = [&](const uint16_t* base, int32_t ld, int32_t cc, uint32_t (&o)[MAXU], bool keep_old) {
        const uint16_t* ub = base + (int64_t)b * HpWp * ld;          // uniform
        const uint32_t loff = (uint32_t)(((rl0 + 1) * Wp + xl + 1) * ld + cc);
        const int64_t sstep = (int64_t)RL * Wp * ld;                  // uniform
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const uint16_t* pk = ub + (k * RL < H ? k * sstep : 0);      // (RL divides H: the unit exists for every row lane or for none)
            const uint32_t t = *reinterpret_cast<const uint32_t*>(pk + loff);
            o[k] = keep_old ? o[k] : t;
        }
    };
    if (!g_straddle) {
        if (g_second) load_all(x2, ld2, c - Ca, v, false); else load_all(x, ld1, c, v, false);
    } else {
        load_all(x, ld1, c < Ca ? c : 0, v, false);
        load_all(x2, ld2, c >= Ca ? c - Ca : 0, v, c < Ca);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        const float t = sm_up((uint16_t)v[k], bf) + sm_up((uint16_t)(v[k] >> 16), bf);
        s += (mine && k * RL < H) ? t : 0.f;
    }
    const float inv_n = 1.0f / (float)((int64_t)HW * cpg);
    const float mean = sm_block_sum(s, s_red, nwaves) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        const float a = sm_up((uint16_t)v[k], bf) - mean, d = sm_up((uint16_t)(v[k] >> 16), bf) - mean;
        q += (mine && k * RL < H) ? fmaf(a, a, d * d) : 0.f;
    }
    const float rstd = rsqrtf(sm_block_sum(q, s_red, nwaves) * inv_n + eps);
    if (!mine) return;
    const float g0 = sm_up(gamma[c], bf) * rstd, g1 = sm_up(gamma[c + 1], bf) * rstd;
    const float h0 = sm_up(beta[c], bf) - mean * g0, h1 = sm_up(beta[c + 1], bf) - mean * g1;
    uint16_t* ybase = y + (tokens ? (int64_t)b * HW : (int64_t)b * HpWp) * C;       // uniform
    const uint32_t doff = (uint32_t)((tokens ? rl * W + xl : (rl + 1) * Wp + xl + 1) * C + c);
    const int64_t dstep = (int64_t)RL * (tokens ? W : Wp) * C;       // uniform
#pragma unroll
    for (int k = 0; k < MAXU; ++k) {
        float t0 = fmaf(sm_up((uint16_t)v[k], bf), g0, h0), t1 = fmaf(sm_up((uint16_t)(v[k] >> 16), bf), g1, h1);
        if (act) {
            t0 = t0 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t0));
            t1 = t1 * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t1));
        }
        if (k * RL < H) *reinterpret_cast<uint32_t*>(ybase + k * dstep + doff) = (uint32_t)sm_cvt(t0, bf) | ((uint32_t)sm_cvt(t1, bf) << 16);
    }
    if (!tokens) {                                                    // the border rows of the PF output, this group's channel slice
        const int32_t nb = 2 * Wp + 2 * H, PS = W * RL;
        uint16_t* yb = y + c + (int64_t)b * HpWp * C;
        for (int32_t j = pl; j < nb; j += PS) {
            int32_t row;
            if (j < Wp) row = j;
            else if (j < 2 * Wp) row = (Hp - 1) * Wp + (j - Wp);
            else { const int32_t k = j - 2 * Wp; row = (1 + (k >> 1)) * Wp + ((k & 1) ? Wp - 1 : 0); }
            *reinterpret_cast<uint32_t*>(yb + (int64_t)row * C) = 0u;
        }
    }
}

// out[b, :] = table[clamp(index[b * idx_stride]), :], 16 bytes per thread
__global__ __launch_bounds__(256) void gsw_gather_rows_kernel(const uint4* __restrict__ table, int64_t ld16, int64_t nrows, const int64_t* __restrict__ index, int32_t idx_stride,
                                                             uint4* __restrict__ out, int64_t out_ld16, int32_t n16) {
    const int32_t b = blockIdx.y;
    int64_t r = index[(int64_t)b * idx_stride];
    r = r < 0 ? 0 : r >= nrows ? nrows - 1 : r;
    for (int32_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) out[(int64_t)b * out_ld16 + i] = table[r * ld16 + i];
}

// x [B, Cin, H, W] -> PF rows [B, H+2, W+2, Cp] (Cp % 8 == 0): zero border, channels >= Cin zero.  One thread per (row, 8-channel vector).
__global__ __launch_bounds__(256) void gsw_nchw_to_pf_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int32_t B, int32_t Cin, int32_t H, int32_t W, int32_t Cp) {
    const int32_t Hp = H + 2, Wp = W + 2, cv = Cp >> 3;
    const int64_t total = (int64_t)B * Hp * Wp * cv;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int32_t cvec = (int32_t)(u % cv);
        const int64_t row = u / cv;
        const int32_t xx = (int32_t)(row % Wp), yy = (int32_t)((row / Wp) % Hp), b = (int32_t)(row / ((int64_t)Hp * Wp));
        uint16_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int32_t ch = cvec * 8 + k;
                if (ch < Cin) h[k] = x[(((int64_t)b * Cin + ch) * H + (yy - 1)) * W + (xx - 1)];
            }
        }
        *reinterpret_cast<uint4*>(y + row * Cp + cvec * 8) = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                                                       (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
    }
}

typedef _Float16 sm_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 sm_b8 __attribute__((ext_vector_type(8)));
typedef float sm_f4 __attribute__((ext_vector_type(4)));

// 3x3 stride-1 convolution PF [B, H, W, C] -> NCHW [B, Nout <= 16, H, W] (+ bias): the UNet's conv_out (320 -> 4).  One WORKGROUP per 16 consecutive
// interior pixels, its four waves splitting the 9 C / 32 K steps among them (wave w: steps w, w + 4, ...: a 64 x 64 image is 256 workgroups of
// ~23 dependent steps instead of 64 of 90); v_mfma_f32_16x16x32 with the weights as the A operand (rows >= Nout are zero lanes) and the pixels'
// tap-shifted PF rows as the B operand, both straight from global memory (the weights are 9 C Nout * 2 bytes = 23 KB: L1 / L2 hits), SDEPTH steps of
// loads in flight; the partial sums meet in LDS and are added in wave order.  Lanes 0..15 end up with outputs n = 0..3 (4..7 in lanes 16..31, ...) of
// pixel lane & 15.  w: [Nout][9 * C] tap-major, channel-minor.
template <typename FRAG, bool BF>
__global__ __launch_bounds__(256) void gsw_conv3x3_pf_nchw_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
                                                                 uint16_t* __restrict__ y, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Nout) {
    constexpr int SDEPTH = 8;
    __shared__ sm_f4 s_part[3][64];
    const int32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t HW = H * W, Wp = W + 2, HpWp = (H + 2) * Wp;
    const int64_t tile = blockIdx.x;                                   // 16-pixel tile
    const int64_t npix = (int64_t)B * HW;
    const int32_t li = lane & 15, q = lane >> 4;
    int64_t pix = tile * 16 + li;
    if (pix >= npix) pix = npix - 1;
    const int32_t b = (int32_t)(pix / HW), pp = (int32_t)(pix - (int64_t)b * HW), yy = pp / W, xx = pp - yy * W;
    const uint16_t* xrow = x + ((int64_t)b * HpWp + (int64_t)(yy + 1) * Wp + xx + 1) * C + q * 8;      // centre tap, this lane's k-chunk
    const bool wlane = li < Nout;
    const uint16_t* wrow = w + (int64_t)(wlane ? li : 0) * 9 * C + q * 8;
    const int32_t cb = C >> 5, nsteps = 9 * cb;                        // 32 k-values per step
    sm_f4 acc = sm_f4{0.f, 0.f, 0.f, 0.f};
    for (int32_t s0 = wave; s0 < nsteps; s0 += 4 * SDEPTH) {
        FRAG fa[SDEPTH], fb[SDEPTH];
#pragma unroll
        for (int j = 0; j < SDEPTH; ++j) {
            const int32_t st = s0 + 4 * j;
            const bool ok = st < nsteps;                               // (wave-uniform)
            const int32_t tap = ok ? st / cb : 0, ch = ok ? st - tap * cb : 0;
            const int32_t toff = ((tap / 3) - 1) * Wp + (tap % 3) - 1;
            const uint4 ub = *reinterpret_cast<const uint4*>(xrow + (int64_t)toff * C + ch * 32);
            uint4 ua = make_uint4(0, 0, 0, 0);
            if (wlane && ok) ua = *reinterpret_cast<const uint4*>(wrow + (int64_t)tap * C + ch * 32);
            fa[j] = __builtin_bit_cast(FRAG, ua);                      // steps past the end multiply zero weights
            fb[j] = __builtin_bit_cast(FRAG, ub);
        }
#pragma unroll
        for (int j = 0; j < SDEPTH; ++j) {
            if constexpr (BF) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j], fb[j], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[j], fb[j], acc, 0, 0, 0);
        }
    }
    if (wave > 0) s_part[wave - 1][lane] = acc;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) acc += s_part[k][lane];
    // D[n = 4 q + j][pixel li]
    if (tile * 16 + li < npix) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t n = 4 * q + j;
            if (n < Nout) {
                const float bv = bias ? sm_up(bias[n], BF) : 0.f;
                y[((int64_t)b * Nout + n) * HW + pp] = sm_cvt(acc[j] + bv, BF);
            }
        }
    }
}

}  // namespace

int gsw_groupnorm_pf_fused(const void* x_dev, const void* x2_dev, int Ca, const void* gamma_dev, const void* beta_dev, void* out_dev, int B, int H, int W,
                           int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream) {
    if (x2_dev && (Ca <= 0 || Ca >= C || (Ca & 7))) return GSW_ERR_BAD_ARG;
    if (!x_dev || !gamma_dev || !beta_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if ((C & 7) || C % groups || ((C / groups) & 1) || B > 65535 || (int64_t)B * (H + 2) * (W + 2) * C >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    const int npair = C / groups / 2;
    // threads: (pair, column, row lane) -- whole image rows per row lane, as many row lanes as 1024 threads allow
    if ((int64_t)npair * W > 1024) return GSW_ERR_UNSUPPORTED;        // one image row per row lane does not fit a workgroup: use gsw_groupnorm_pf2
    int RL = std::min(1024 / (npair * W), H);
    while (H % RL) --RL;                                              // a divisor of H: every row lane owns the same number of rows
    const int threads = std::min(1024, (npair * W * RL + 63) / 64 * 64);
    const int units = (H + RL - 1) / RL;
    if (units > 32) return GSW_ERR_UNSUPPORTED;                       // the group does not fit the registers of one workgroup (128 per thread at 1024 threads): use gsw_groupnorm_pf2
    hipStream_t st = (hipStream_t)stream;
    const int bf = dtype == GSW_BF16;
#define GSW_GNF_LAUNCH(MU)                                                                                                                        \
    hipLaunchKernelGGL(gsw_gn_fused_kernel<MU>, dim3(groups, B), dim3(threads), 0, st, (const uint16_t*)x_dev, (const uint16_t*)x2_dev, Ca, (const uint16_t*)gamma_dev, \
                       (const uint16_t*)beta_dev, (uint16_t*)out_dev, C, groups, H, W, npair, RL, eps, act, out_tokens, bf)
    if (units <= 8) GSW_GNF_LAUNCH(8); else if (units <= 16) GSW_GNF_LAUNCH(16); else GSW_GNF_LAUNCH(32);
#undef GSW_GNF_LAUNCH
    GSW_SM_CHECK_LAUNCH();
    return GSW_OK;
}

int gsw_gather_rows(const void* table_dev, int64_t ld_bytes, int64_t nrows, const int64_t* index_dev, int index_stride, void* out_dev, int64_t out_ld_bytes,
                    int B, int64_t row_bytes, void* stream) {
    if (!table_dev || !index_dev || !out_dev || B <= 0 || B > 65535 || nrows <= 0 || row_bytes <= 0 || index_stride < 0) return GSW_ERR_BAD_ARG;
    if ((row_bytes & 15) || (ld_bytes & 15) || (out_ld_bytes & 15) || ld_bytes < row_bytes || out_ld_bytes < row_bytes || (((uintptr_t)table_dev | (uintptr_t)out_dev) & 15)
        || row_bytes > ((int64_t)1 << 34)) return GSW_ERR_BAD_ARG;
    const int32_t n16 = (int32_t)(row_bytes >> 4);
    hipLaunchKernelGGL(gsw_gather_rows_kernel, dim3((uint32_t)std::min<int64_t>((n16 + 255) / 256, 1024), B), dim3(256), 0, (hipStream_t)stream, (const uint4*)table_dev, ld_bytes >> 4, nrows,
                       index_dev, index_stride, (uint4*)out_dev, out_ld_bytes >> 4, n16);
    GSW_SM_CHECK_LAUNCH();
    return GSW_OK;
}

int gsw_nchw_to_pf(const void* x_dev, void* y_dev, int B, int Cin, int H, int W, int Cp, int dtype, void* stream) {
    if (!x_dev || !y_dev || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cp < Cin) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if ((Cp & 7) || ((uintptr_t)y_dev & 15)) return GSW_ERR_UNSUPPORTED;
    const int64_t total = (int64_t)B * (H + 2) * (W + 2) * (Cp >> 3);
    hipLaunchKernelGGL(gsw_nchw_to_pf_kernel, dim3((uint32_t)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_dev, (uint16_t*)y_dev,
                       B, Cin, H, W, Cp);
    GSW_SM_CHECK_LAUNCH();
    return GSW_OK;
}

int gsw_conv3x3_pf_nchw(const void* x_dev, const void* w_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int Nout, int dtype, void* stream) {
    if (!x_dev || !w_dev || !y_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || Nout <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if ((C & 31) || Nout > 16 || (((uintptr_t)x_dev | (uintptr_t)w_dev) & 15) || (int64_t)B * (H + 2) * (W + 2) * C >= ((int64_t)1 << 40)) return GSW_ERR_UNSUPPORTED;
    const uint32_t grid = (uint32_t)(((int64_t)B * H * W + 15) / 16);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GSW_F16)
        hipLaunchKernelGGL((gsw_conv3x3_pf_nchw_kernel<sm_h8, false>), dim3(grid), dim3(256), 0, st, (const uint16_t*)x_dev, (const uint16_t*)w_dev, (const uint16_t*)bias_dev, (uint16_t*)y_dev, B, H, W, C, Nout);
    else
        hipLaunchKernelGGL((gsw_conv3x3_pf_nchw_kernel<sm_b8, true>), dim3(grid), dim3(256), 0, st, (const uint16_t*)x_dev, (const uint16_t*)w_dev, (const uint16_t*)bias_dev, (uint16_t*)y_dev, B, H, W, C, Nout);
    GSW_SM_CHECK_LAUNCH();
    return GSW_OK;
}

// ================================================================================================================================================
// gsw_mm_small_kernel: dense linears of the transformer blocks at SMALL M (one or two images: 64 ... 8192 token rows).  The engine's 128 x 160
// tiles leave most of the chip without a tile there (M = 256, N = 1280: 16 tiles on 256 CUs, 18 us for 0.8 GFLOP) and its split-K needs a second
// launch.  Here a workgroup of four waves owns a (16 TA) x (16 TB) output tile and SPLITS K FOUR WAYS INSIDE the workgroup: wave w multiplies the
// 32-wide K granules w, w + 4, ... with fragments loaded straight from global memory into the MFMA operand layout (row-major rows are 16 contiguous
// bytes per lane; no LDS staging, D granules of loads in flight per wave), the four partial tiles meet in LDS and are added in wave order (a fixed
// order: deterministic), and the epilogue -- distributed over the waves -- does what the engine's epilogues do, with their rounding points: bias,
// residual, GEGLU, transposed output, token -> PF scatter, LayerNorm fold (straight from the producer's raw row records: no finishing launch) and the
// row records for the next LayerNorm.  Weights are the MFMA A operand (a lane ends with 4 consecutive output columns of one row), swapped for the
// transposed output (4 consecutive tokens of one output row).
// ================================================================================================================================================
#include "gswm_mmtypes.h"

namespace {

struct SmArgs {
    const void* x; const void* w; const void* bias; const void* resid; void* y;
    int64_t ldx, ldw, ldr, ldy;
    int32_t M, K, N, mode;            // mode: GSW_GEMM_PLAIN / GSW_GEMM_GEGLU / GSW_GEMM_TRANS / GSW_GEMM_TOK2PF
    int32_t S, Wimg;                  // TRANS: tokens per image; TOK2PF: tokens per image, image width
    const float* ln_rec;              // LayerNorm fold: raw row records of x, [M][ln_slots][2] (sum, sum of squares), or null
    int32_t ln_slots; float ln_eps, ln_inv_c;
    const float* ln_u; const float* ln_v;
    float* rowstats;                  // out, or null: [M][tiles_n][2] (sum, sum of squares) of the stored values per row and column tile
    int32_t tiles_m, tiles_n;
};

template <typename T, int TA, int TB, bool SWAP>
__global__ __launch_bounds__(256) void gsw_mm_small_kernel(const SmArgs p) {
    constexpr int D = 4, NACC = TA * TB, TM = 16 * TA, TN = 16 * TB;
    typedef typename MM<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) float sm_lds[];
    mm_f4* s_acc = reinterpret_cast<mm_f4*>(sm_lds);                 // [4 waves][NACC][64 lanes]
    float* s_ln = sm_lds + 4 * NACC * 256;                           // [TM][2]: (rstd, -rstd mean) of the tile's rows
    float* s_rs = s_ln + TM * 2;                                     // [NACC][16][2]: per accumulator and row the (sum, sum of squares) of its 16 columns
    const int32_t lane = threadIdx.x & 63;
    const int32_t wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int32_t li = lane & 15, q = lane >> 4;
    const int32_t tm = (int32_t)blockIdx.x % p.tiles_m, tn = (int32_t)blockIdx.x / p.tiles_m;
    const int32_t m0 = tm * TM, n0 = tn * TN;

    if (p.ln_rec && (int32_t)threadIdx.x < TM) {                     // (published by the barrier in front of the epilogue)
        const int32_t m = m0 + (int32_t)threadIdx.x;
        float sm = 0.f, sq = 0.f;
        if (m < p.M) {
            const float2* r = reinterpret_cast<const float2*>(p.ln_rec) + (int64_t)m * p.ln_slots;
            for (int32_t k = 0; k < p.ln_slots; ++k) { const float2 t = r[k]; sm += t.x; sq += t.y; }
        }
        const float mean = sm * p.ln_inv_c;
        const float rstd = rsqrtf(fmaxf(sq * p.ln_inv_c - mean * mean, 0.f) + p.ln_eps);
        s_ln[2 * threadIdx.x] = rstd;
        s_ln[2 * threadIdx.x + 1] = -rstd * mean;
    }

    const T* X = reinterpret_cast<const T*>(p.x);
    const T* W = reinterpret_cast<const T*>(p.w);
    const uint32_t xoff = (uint32_t)(li * (int32_t)p.ldx + 8 * q), woff = (uint32_t)(li * (int32_t)p.ldw + 8 * q);
    // (wave-uniform row-tile base) + (32-bit lane offset): row tiles past M / N re-read the tile's first one and are never stored
    auto load_granule = [&](int32_t g, frag (&fa)[TA], frag (&fb)[TB]) {
        const int64_t k0 = (int64_t)g * 32;
#pragma unroll
        for (int a = 0; a < TA; ++a) {
            const int32_t r = m0 + 16 * a < p.M ? m0 + 16 * a : m0;
            fa[a] = *reinterpret_cast<const frag*>(X + ((int64_t)r * p.ldx + k0) + xoff);
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            const int32_t r = n0 + 16 * b < p.N ? n0 + 16 * b : n0;
            fb[b] = *reinterpret_cast<const frag*>(W + ((int64_t)r * p.ldw + k0) + woff);
        }
    };
    mm_f4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = mm_f4{0.f, 0.f, 0.f, 0.f};
    const int32_t ng = p.K >> 5;
    const int32_t ngw = (ng - wave + 3) >> 2;                         // this wave's granules: wave + 4 i, i < ngw (the host guarantees ng >= 4)
    frag fa[D][TA], fb[D][TB];
#pragma unroll
    for (int d = 0; d < D; ++d) load_granule(wave + 4 * (d < ngw ? d : 0), fa[d], fb[d]);
    for (int32_t i = 0; i < ngw; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (i + d < ngw) {
#pragma unroll
                for (int a = 0; a < TA; ++a)
#pragma unroll
                    for (int b = 0; b < TB; ++b)
                        acc[a * TB + b] = SWAP ? MM<T>::mma(fa[d][a], fb[d][b], acc[a * TB + b]) : MM<T>::mma(fb[d][b], fa[d][a], acc[a * TB + b]);
            }
            const int32_t nx = i + d + D;
            load_granule(wave + 4 * (nx < ngw ? nx : 0), fa[d], fb[d]);      // (past the end: a cached granule, its values are never multiplied)
        }
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s_acc[(wave * NACC + i) * 64 + lane] = acc[i];
    __syncthreads();

    const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
    const uint16_t* resid = reinterpret_cast<const uint16_t*>(p.resid);
    uint16_t* Y = reinterpret_cast<uint16_t*>(p.y);
    const bool want_rs = p.rowstats != nullptr;
    for (int32_t idx = wave; idx < NACC; idx += 4) {
        const int32_t ta = idx / TB, tb = idx - ta * TB;
        mm_f4 a = s_acc[idx * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) a += s_acc[(w * NACC + idx) * 64 + lane];
        const bool live = m0 + 16 * ta < p.M && n0 + 16 * tb < p.N;   // (wave-uniform)
        float rs_s = 0.f, rs_q = 0.f;
        if (live) {
            if constexpr (!SWAP) {
                const int32_t m = m0 + 16 * ta + li, n = n0 + 16 * tb + 4 * q;
                float o[4];
                if (p.ln_rec) {
                    const float rstd = s_ln[2 * (16 * ta + li)], nrm = s_ln[2 * (16 * ta + li) + 1];
                    const mm_f4 u = *reinterpret_cast<const mm_f4*>(p.ln_u + n), v = *reinterpret_cast<const mm_f4*>(p.ln_v + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j], rstd, fmaf(nrm, u[j], v[j]));
                } else {
                    uint2 bq = make_uint2(0, 0);
                    if (bias) bq = *reinterpret_cast<const uint2*>(bias + n);
                    o[0] = a[0] + MM<T>::up_lo(bq.x); o[1] = a[1] + MM<T>::up_hi(bq.x); o[2] = a[2] + MM<T>::up_lo(bq.y); o[3] = a[3] + MM<T>::up_hi(bq.y);
                }
                uint32_t h0 = MM<T>::cvt2(o[0], o[1]), h1 = MM<T>::cvt2(o[2], o[3]);
                if (p.mode == GSW_GEMM_GEGLU) {
                    // packed weight rows: [8 values | 8 gates] per 16-row block -> lanes q < 2 hold the values of outputs 4 q .. 4 q + 3, lanes q + 2 their gates
                    const uint32_t g0 = (uint32_t)__shfl_xor((int)h0, 32, 64), g1 = (uint32_t)__shfl_xor((int)h1, 32, 64);
                    if (q < 2) {
                        const uint32_t e0 = MM<T>::cvt2(mm_gelu(MM<T>::up_lo(g0)), mm_gelu(MM<T>::up_hi(g0))), e1 = MM<T>::cvt2(mm_gelu(MM<T>::up_lo(g1)), mm_gelu(MM<T>::up_hi(g1)));
                        *reinterpret_cast<uint2*>(Y + (int64_t)m * p.ldy + ((n0 + 16 * tb) >> 1) + 4 * q) = make_uint2(MM<T>::mul2(h0, e0), MM<T>::mul2(h1, e1));
                    }
                } else {
                    int64_t orow = m;
                    if (p.mode == GSW_GEMM_TOK2PF) {                   // token (b, y, x) -> interior row of the PF tensor [B, H+2, W+2]
                        const int32_t b = m / p.S, ii = m - b * p.S, yy = ii / p.Wimg, xx = ii - yy * p.Wimg;
                        const int32_t Wp = p.Wimg + 2, Hp = p.S / p.Wimg + 2;
                        orow = (int64_t)b * Hp * Wp + (int64_t)(yy + 1) * Wp + (xx + 1);
                    }
                    if (resid) {
                        const uint2 r = *reinterpret_cast<const uint2*>(resid + orow * p.ldr + n);
                        h0 = MM<T>::add2(h0, r.x); h1 = MM<T>::add2(h1, r.y);
                    }
                    *reinterpret_cast<uint2*>(Y + orow * p.ldy + n) = make_uint2(h0, h1);
                    if (want_rs) { MM<T>::stat2(h0, rs_s, rs_q); MM<T>::stat2(h1, rs_s, rs_q); }
                }
            } else {
                // transposed output Y[image][n][token]: lane (q, li) holds tokens 4 q .. 4 q + 3 of output row li of the accumulator
                const int32_t m = m0 + 16 * ta + 4 * q, n = n0 + 16 * tb + li;
                float o[4];
                if (p.ln_rec) {
                    const float u = p.ln_u[n], v = p.ln_v[n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j], s_ln[2 * (16 * ta + 4 * q + j)], fmaf(s_ln[2 * (16 * ta + 4 * q + j) + 1], u, v));
                } else {
                    const float bv = bias ? MM<T>::up(bias[n]) : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = a[j] + bv;
                }
                const int32_t b = m / p.S, sidx = m - b * p.S;
                *reinterpret_cast<uint2*>(Y + ((int64_t)b * p.N + n) * p.S + sidx) = make_uint2(MM<T>::cvt2(o[0], o[1]), MM<T>::cvt2(o[2], o[3]));
            }
        }
        if (want_rs) {                                                 // fold the four column chunks of a row (lanes li, li + 16, li + 32, li + 48)
            rs_s += __shfl_xor(rs_s, 16, 64); rs_q += __shfl_xor(rs_q, 16, 64);
            rs_s += __shfl_xor(rs_s, 32, 64); rs_q += __shfl_xor(rs_q, 32, 64);
            if (q == 0) { s_rs[(idx * 16 + li) * 2] = rs_s; s_rs[(idx * 16 + li) * 2 + 1] = rs_q; }
        }
    }
    if (want_rs) {
        __syncthreads();
        if ((int32_t)threadIdx.x < TM) {                               // one thread per row: the column tiles of the row in order
            const int32_t ta = (int32_t)threadIdx.x >> 4, r = (int32_t)threadIdx.x & 15, m = m0 + (int32_t)threadIdx.x;
            if (m < p.M) {
                float s = 0.f, qq = 0.f;
#pragma unroll
                for (int b = 0; b < TB; ++b) { s += s_rs[((ta * TB + b) * 16 + r) * 2]; qq += s_rs[((ta * TB + b) * 16 + r) * 2 + 1]; }
                *reinterpret_cast<float2*>(p.rowstats + ((int64_t)m * p.tiles_n + tn) * 2) = make_float2(s, qq);
            }
        }
    }
}

template <typename T, int TA, int TB>
int sm_launch(const SmArgs& a, hipStream_t st) {
    constexpr size_t ldsb = (size_t)(4 * TA * TB * 256 + 16 * TA * 2 + TA * TB * 32) * sizeof(float);
    static bool attr_done[2] = {false, false};
    const bool swap = a.mode == GSW_GEMM_TRANS;
    if (!attr_done[swap]) {
        hipError_t e = swap ? hipFuncSetAttribute((const void*)gsw_mm_small_kernel<T, TA, TB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)
                            : hipFuncSetAttribute((const void*)gsw_mm_small_kernel<T, TA, TB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        if (e != hipSuccess) return (int)e;
        attr_done[swap] = true;
    }
    const uint32_t grid = (uint32_t)((int64_t)a.tiles_m * a.tiles_n);
    if (swap) hipLaunchKernelGGL((gsw_mm_small_kernel<T, TA, TB, true>), dim3(grid), dim3(256), ldsb, st, a);
    else hipLaunchKernelGGL((gsw_mm_small_kernel<T, TA, TB, false>), dim3(grid), dim3(256), ldsb, st, a);
    return (int)hipGetLastError();
}

template <typename T>
int sm_launch_t(SmArgs& a, int cfg, hipStream_t st) {
    const int ta = cfg == 0 ? 4 : cfg == 1 ? 2 : 1, tb = cfg == 3 ? 2 : 4;
    a.tiles_m = (a.M + 16 * ta - 1) / (16 * ta);
    a.tiles_n = (a.N + 16 * tb - 1) / (16 * tb);
    switch (cfg) {
        case 0: return sm_launch<T, 4, 4>(a, st);
        case 1: return sm_launch<T, 2, 4>(a, st);
        case 2: return sm_launch<T, 1, 4>(a, st);
        default: return sm_launch<T, 1, 2>(a, st);
    }
}

}  // namespace

// Tile of the small-M kernel for a shape: the largest of 64 x 64, 32 x 64, 16 x 64, 16 x 32 that still gives ~200 workgroups (or the smallest).
// Returns the configuration index, or -1 when the shape is not this kernel's (then the engine runs it).
int gsw_gemm_small_config(int64_t M, int K, int N, int mode) {
    if (M <= 0 || K < 128 || (K & 31) || (N & 15) || (M & 15) || M > 16384) return -1;
    if (mode == GSW_GEMM_GEGLU && (N & 31)) return -1;
    static const int ta[4] = {4, 2, 1, 1}, tb[4] = {4, 4, 4, 2};
    for (int c = 0; c < 4; ++c) {
        const int64_t wgs = ((M + 16 * ta[c] - 1) / (16 * ta[c])) * ((N + 16 * tb[c] - 1) / (16 * tb[c]));
        if (wgs >= 192 || c == 3) return c;
    }
    return 3;
}

int gsw_gemm_small(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr, void* y_dev, int64_t ldy,
                   int64_t M, int K, int N, int mode, int S, int Wimg, const float* ln_records_dev, int ln_slots, float ln_eps, const float* ln_u_dev,
                   const float* ln_v_dev, float* rowstats_dev, int64_t rowstats_capacity, int* rowstats_slots, int config, int dtype, void* stream) {
    if (rowstats_slots) *rowstats_slots = 0;
    if (!x_dev || !w_dev || !y_dev || M <= 0 || K <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (mode != GSW_GEMM_PLAIN && mode != GSW_GEMM_GEGLU && mode != GSW_GEMM_TRANS && mode != GSW_GEMM_TOK2PF) return GSW_ERR_BAD_ARG;
    const int cfg = config >= 0 ? config : gsw_gemm_small_config(M, K, N, mode);
    if (cfg < 0 || cfg > 3 || gsw_gemm_small_config(M, K, N, mode) < 0) return GSW_ERR_UNSUPPORTED;
    if (ldx < K || ldw < K || (ldx & 7) || (ldw & 7) || (ldy & 3) || (ldr & 3) || M * ldx >= ((int64_t)1 << 31) || (int64_t)N * ldw >= ((int64_t)1 << 31)
        || ((uintptr_t)x_dev & 15) || ((uintptr_t)w_dev & 15) || ((uintptr_t)y_dev & 7) || ((uintptr_t)bias_dev & 7) || ((uintptr_t)resid_dev & 7)) return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_GEGLU && resid_dev) return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_TRANS && (resid_dev || S <= 0 || (S & 3) || M % S)) return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_TOK2PF && (S <= 0 || Wimg <= 0 || S % Wimg || M % S)) return GSW_ERR_BAD_ARG;
    const int64_t ncols = mode == GSW_GEMM_GEGLU ? N / 2 : N;
    if (mode != GSW_GEMM_TRANS && (ldy < ncols || (resid_dev && ldr < ncols))) return GSW_ERR_BAD_ARG;
    if (ln_records_dev && (ln_slots <= 0 || !ln_u_dev || !ln_v_dev || bias_dev || mode == GSW_GEMM_TOK2PF || (((uintptr_t)ln_u_dev | (uintptr_t)ln_v_dev) & 15) || ((uintptr_t)ln_records_dev & 7)))
        return GSW_ERR_BAD_ARG;
    SmArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.resid = resid_dev; a.y = y_dev;
    a.ldx = ldx; a.ldw = ldw; a.ldr = ldr; a.ldy = ldy;
    a.M = (int32_t)M; a.K = K; a.N = N; a.mode = mode; a.S = S > 0 ? S : 1; a.Wimg = Wimg > 0 ? Wimg : 1;
    a.ln_rec = ln_records_dev; a.ln_slots = ln_slots; a.ln_eps = ln_eps; a.ln_inv_c = 1.0f / (float)K; a.ln_u = ln_u_dev; a.ln_v = ln_v_dev;
    a.rowstats = nullptr;
    const int tbn = cfg == 3 ? 2 : 4;
    const int64_t tiles_n = (N + 16 * tbn - 1) / (16 * tbn);
    if (rowstats_dev && mode == GSW_GEMM_PLAIN && M * tiles_n * 2 <= rowstats_capacity && !((uintptr_t)rowstats_dev & 7)) {
        a.rowstats = rowstats_dev;
        if (rowstats_slots) *rowstats_slots = (int)tiles_n;
    }
    const int e = dtype == GSW_F16 ? sm_launch_t<_Float16>(a, cfg, (hipStream_t)stream) : sm_launch_t<__bf16>(a, cfg, (hipStream_t)stream);
    if (e != 0) { g_last_hip_error = e; return GSW_ERR_HIP; }
    return GSW_OK;
}
