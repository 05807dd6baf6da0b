// gswm_conv.hip -- convolution front end of the eps model (rows X2 / G1 of SURVEY.md section 8a), GroupNorm / LayerNorm kernels.  gfx950 only.
//
// Activation format "padded-flat NHWC" (PF): an image batch [B, H, W, C] is stored with a one-pixel zero border as a 2-D
// matrix X[(b, y, x) -> b*Hp*Wp + y*Wp + x][C] (Hp = H+2, Wp = W+2), plus G = Wp+1 guard rows of zeros in front and behind.
// In that domain a 3x3 tap is a CONSTANT row offset ((kh-1)*Wp + (kw-1)), so the convolution is one GEMM
//     Y[m, n] = sum_t sum_c X[row(m) + off_t, c] * Wt[n, t*C + c]  (+ bias[n] + rowbias[b(m), n] + residual[m, n])
// whose A-tile loader is a plain strided copy: no im2col buffer, no boundary tests in the K loop (border pixels read the
// zero padding).  Border rows of Y are written as zeros so the result is again a valid PF tensor.  1x1 convolutions, the 2x2 sub-pixel
// form of Upsample2D and stride-2 downsampling are the same GEMM with a different tap table / row map.
//
// Every convolution with >= 128 output channels is translated here into the matmul engine's argument block (csrc/gswm_mm.hip: taps =
// K runs of a segment, conv_shortcut over cat(x, skip) = two more segments).  What remains in this file as a kernel is the 64-column
// tile for the 4-channel edges (conv_in / conv_out / VAE 3-channel and 8-channel ends padded to one 64-wide tile, 0.3 % of a run):
// BM = 256 pixels x BN = 64 channels x BK = 64, 256 threads = 4 waves, each wave a 64 x 64 sub-tile as 2 x 2 v_mfma_f32_32x32x16 tiles
// (weights as the A operand, activations as the B operand, so a lane ends up with 4 consecutive output channels of one pixel).
// Operands are staged with global_load_lds (16 B per lane, 1 KiB per wave instruction) into an LDS image whose 16-byte chunks are
// XOR-swizzled with (row >> 1) & 7 on the SOURCE address, which makes every ds_read_b128 fragment read conflict-free.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <algorithm>
#include <stdlib.h>

#include "../../include/gswm.h"
#include "gswm_mm.h"

typedef _Float16 gsw_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 gsw_b8 __attribute__((ext_vector_type(8)));
typedef float gsw_f16v __attribute__((ext_vector_type(16)));

#define CV_BM 256
#define CV_BN 64
#define CV_BK 64
#define CV_THREADS 256
#define CV_OUT_STRIDE 136   // bytes per row of the epilogue image (128 + 8: conflict-free ds_write_b64, 8-byte aligned reads)

struct ConvArgs {
    const void* x;        // PF activations, pointer to row 0 (guards live at negative rows)
    const void* w;        // [N][T*C] K-contiguous weights
    const void* bias;     // [N] or null
    const void* rowbias;  // [B][ldrb] or null (time-embedding projection; rows ldrb >= N elements apart)
    const void* resid;    // [M][N] or null (PF, same geometry as the output)
    void* y;              // [M][N] PF output
    int32_t tap_off[9];   // row offset of each tap in the INPUT PF domain
    int32_t ntaps;
    int32_t C, N;         // input / output channels
    int32_t M;            // output rows = B * Hp * Wp (output geometry)
    int32_t Hp, Wp;       // output padded geometry
    int32_t in_Hp, in_Wp; // input padded geometry
    int32_t stride;       // 1 or 2
    int32_t ldrb;         // row stride of rowbias in elements
    int32_t ldx;          // row stride of x in elements (>= C; lets the input be a channel slice of a wider tensor)
    int32_t dense;        // 1: plain GEMM on a dense [M, C] matrix (no border rows, no row map)
    int32_t geglu;        // (unused since the dense linears moved to the matmul engine)
    int32_t ldy;          // row stride of y in elements (N, or N/2 with geglu)
    int32_t up;           // engine only: 0, or 1 + dy*2 + dx = this launch computes output parity (dy, dx) of a 2x nearest-neighbour
                          // upsample + 3x3 convolution from the LOW-resolution input (sub-pixel decomposition, ntaps == 4)
    // engine only: up to two extra 1x1 operand segments appended to the K loop (the resnet's conv_shortcut folded in,
    // its concatenated input given as two tensors): K = 9*C + C1 + C2, weights [N][9*C | C1 | C2]
    const void* x1; const void* x2;
    int32_t C1, C2;
};

template <typename T> struct Mfma;
template <> struct Mfma<_Float16> {
    typedef gsw_h8 frag;
    static __device__ __forceinline__ gsw_f16v mma(frag a, frag b, gsw_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { return __half_as_ushort(__float2half_rn(f)); }
    static __device__ __forceinline__ float up(uint16_t h) { return __half2float(__ushort_as_half(h)); }
};
template <> struct Mfma<__bf16> {
    typedef gsw_b8 frag;
    static __device__ __forceinline__ gsw_f16v mma(frag a, frag b, gsw_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        union { __hip_bfloat16 h; uint16_t u; } c; c.h = __float2bfloat16(f); return c.u;
    }
    static __device__ __forceinline__ float up(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
};

template <typename T>
__global__ __launch_bounds__(CV_THREADS) void gsw_conv_gemm_kernel(ConvArgs p) {
    // one LDS array (a second __shared__ object would make hipcc drain vmcnt before every ds_read, cdna guide 5/4a)
    __shared__ __attribute__((aligned(16))) uint8_t lds[(CV_BM + CV_BN) * CV_BK * 2];
    uint8_t* ldsX = lds;                         // [256 rows][128 B], chunk-swizzled
    uint8_t* ldsW = lds + CV_BM * CV_BK * 2;     // [64 rows][128 B]
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // XCD-aware tile order: hardware block b runs on XCD b % 8 (its own L2).  Give every XCD a contiguous range of logical
    // tiles (bijective remap, cdna guide T1), N tiles fastest, so the workgroups that share an activation tile run back to
    // back on the SAME XCD and find it in that L2 instead of each pulling it from HBM.
    const uint32_t nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7u;
    const uint32_t xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    const uint32_t logical = (xcd < r8 ? xcd * (q8 + 1u) : r8 * (q8 + 1u) + (xcd - r8) * q8) + idx;
    const uint32_t ntn = (uint32_t)p.N / CV_BN;
    const uint32_t tile_n = logical % ntn, tile_m = logical / ntn;
    const int32_t m0 = (int32_t)tile_m * CV_BM, n0 = (int32_t)tile_n * CV_BN;
    const int32_t HpWp = p.Hp * p.Wp;

    // per-lane source rows of the 8 activation loads this wave issues per K block (input-domain row of output row m)
    int32_t xrow[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int32_t r = (int32_t)wave * 64 + i * 8 + (int32_t)(lane >> 3);
        int32_t m = m0 + r;
        if (m >= p.M) m = p.M - 1;
        int32_t src = m;
        if (p.stride == 2) {   // output pixel (yo, xo) reads input rows (2yo - 1 + kh) -> padded (2yo + kh) = base + kh*in_Wp + kw
            const int32_t b = m / HpWp, q = m - b * HpWp;
            int32_t yo = q / p.Wp - 1, xo = q - (q / p.Wp) * p.Wp - 1;
            yo = yo < 0 ? 0 : yo; xo = xo < 0 ? 0 : xo;
            const int32_t Ho = p.Hp - 2, Wo = p.Wp - 2;
            yo = yo >= Ho ? Ho - 1 : yo; xo = xo >= Wo ? Wo - 1 : xo;
            src = b * p.in_Hp * p.in_Wp + (2 * yo) * p.in_Wp + 2 * xo;
        }
        xrow[i] = src;
    }
    const uint32_t pc = lane & 7u;               // physical 16-byte chunk this lane fills in its LDS row
    const T* X = reinterpret_cast<const T*>(p.x);
    const T* W = reinterpret_cast<const T*>(p.w);
    const int32_t Ktot = p.ntaps * p.C;

    gsw_f16v acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int32_t kc_per_tap = p.C / CV_BK;
    for (int32_t kc = 0; kc < kc_per_tap; ++kc) {        // channel-block-major, tap-minor (L2 reuse across taps)
        for (int32_t t = 0; t < p.ntaps; ++t) {
            const int32_t off = p.tap_off[t];
            // ---- stage: activations (8 x 1 KiB per wave) and weights (2 x 1 KiB per wave)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t r = wave * 64u + i * 8u + (lane >> 3);
                const uint32_t c = pc ^ ((r >> 1) & 7u);                       // logical chunk stored at physical chunk pc
                const T* src = X + (int64_t)(xrow[i] + off) * p.ldx + kc * CV_BK + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsX + (wave * 64u + i * 8u) * 128u), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t r = wave * 16u + i * 8u + (lane >> 3);
                const uint32_t c = pc ^ ((r >> 1) & 7u);
                const T* src = W + (int64_t)(n0 + (int32_t)r) * Ktot + t * p.C + kc * CV_BK + c * 8;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsW + (wave * 16u + i * 8u) * 128u), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // ---- compute: 4 k-steps x (2 x 2) MFMA tiles
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                typename Mfma<T>::frag wf[2], xf[2];
                const uint32_t lc = (uint32_t)ks * 2u + (lane >> 5);           // logical chunk of this lane's 8 k-values
#pragma unroll
                for (int in = 0; in < 2; ++in) {
                    const uint32_t r = (uint32_t)in * 32u + (lane & 31u);
                    wf[in] = *reinterpret_cast<const typename Mfma<T>::frag*>(ldsW + r * 128u + ((lc ^ ((r >> 1) & 7u)) << 4));
                }
#pragma unroll
                for (int im = 0; im < 2; ++im) {
                    const uint32_t r = wave * 64u + (uint32_t)im * 32u + (lane & 31u);
                    xf[im] = *reinterpret_cast<const typename Mfma<T>::frag*>(ldsX + r * 128u + ((lc ^ ((r >> 1) & 7u)) << 4));
                }
#pragma unroll
                for (int in = 0; in < 2; ++in)
#pragma unroll
                    for (int im = 0; im < 2; ++im) acc[in][im] = Mfma<T>::mma(wf[in], xf[im], acc[in][im]);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: D[n][m] fragments -> LDS image [m][n] (bias added) -> coalesced 16-byte row stores
    // lane: m = wave*64 + im*32 + (lane & 31); n = in*32 + 8*rg + 4*(lane >> 5) + j   (j = 0..3 = reg & 3, rg = reg >> 2)
    const T* bias = reinterpret_cast<const T*>(p.bias);
#pragma unroll
    for (int in = 0; in < 2; ++in)
#pragma unroll
        for (int im = 0; im < 2; ++im) {
            const uint32_t m = wave * 64u + (uint32_t)im * 32u + (lane & 31u);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const uint32_t n = (uint32_t)in * 32u + 8u * rg + 4u * (lane >> 5);
                uint16_t h[4];
                uint2 bw = make_uint2(0, 0);
                if (bias) bw = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(bias) + n0 + n);
                const uint16_t bh[4] = {(uint16_t)bw.x, (uint16_t)(bw.x >> 16), (uint16_t)bw.y, (uint16_t)(bw.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = Mfma<T>::cvt(acc[in][im][rg * 4 + j] + (bias ? Mfma<T>::up(bh[j]) : 0.f));
                uint2 pk;
                pk.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
                pk.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                *reinterpret_cast<uint2*>(lds + m * CV_OUT_STRIDE + n * 2u) = pk;
            }
        }
    __syncthreads();
    const uint16_t* rowbias = reinterpret_cast<const uint16_t*>(p.rowbias);
    const uint16_t* resid = reinterpret_cast<const uint16_t*>(p.resid);
    uint16_t* Y = reinterpret_cast<uint16_t*>(p.y);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t q = tid + 256u * i;
        const uint32_t r = q >> 3, cc = q & 7u;
        const int32_t m = m0 + (int32_t)r;
        if (m >= p.M) continue;
        int32_t b = 0;
        bool border = false;
        if (!p.dense) {
            b = m / HpWp;
            const int32_t qq = m - b * HpWp;
            const int32_t yy = qq / p.Wp, xx = qq - yy * p.Wp;
            border = (yy == 0) | (yy == p.Hp - 1) | (xx == 0) | (xx == p.Wp - 1);
        }
        uint4 o = make_uint4(0, 0, 0, 0);
        if (!border) {
            const uint2 lo = *reinterpret_cast<const uint2*>(lds + r * CV_OUT_STRIDE + cc * 16u);
            const uint2 hi = *reinterpret_cast<const uint2*>(lds + r * CV_OUT_STRIDE + cc * 16u + 8u);
            uint32_t w4[4] = {lo.x, lo.y, hi.x, hi.y};
            if (rowbias || resid) {
                uint4 rb = make_uint4(0, 0, 0, 0), rs = make_uint4(0, 0, 0, 0);
                if (rowbias) rb = *reinterpret_cast<const uint4*>(rowbias + (int64_t)b * p.ldrb + n0 + cc * 8);
                if (resid) rs = *reinterpret_cast<const uint4*>(resid + (int64_t)m * p.N + n0 + cc * 8);
                const uint32_t rbw[4] = {rb.x, rb.y, rb.z, rb.w}, rsw[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a0 = Mfma<T>::up((uint16_t)w4[k]) + Mfma<T>::up((uint16_t)rbw[k]) + Mfma<T>::up((uint16_t)rsw[k]);
                    const float a1 = Mfma<T>::up((uint16_t)(w4[k] >> 16)) + Mfma<T>::up((uint16_t)(rbw[k] >> 16)) + Mfma<T>::up((uint16_t)(rsw[k] >> 16));
                    w4[k] = (uint32_t)Mfma<T>::cvt(a0) | ((uint32_t)Mfma<T>::cvt(a1) << 16);
                }
            }
            o = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *reinterpret_cast<uint4*>(Y + (int64_t)m * p.N + n0 + cc * 8) = o;
    }
}


// ------------------------------------------------------------------------------------------------
// GroupNorm (+SiLU) on PF activations.  Channels of a group are NOT contiguous across pixels in NHWC, so the statistics
// are a two-kernel reduction: (1) every workgroup walks a slab of padded pixels with (C/8) x P threads -- thread = one
// 16-byte channel vector of one pixel lane, fully coalesced rows -- and reduces per-channel partial sums into 32 group
// partials (borders are zero, so they drop out of the sums); (2) the apply kernel folds the slab partials of its image,
// then normalises + activates and writes either a PF tensor (borders zero) or dense tokens [B, H*W, C] for the transformer.
// ------------------------------------------------------------------------------------------------
#define GN_MAX_GROUPS 64

__device__ __forceinline__ void up8h(const uint4 u, float (&v)[8], bool bf) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (bf) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
        else { v[2 * i] = __half2float(__ushort_as_half((uint16_t)w[i])); v[2 * i + 1] = __half2float(__ushort_as_half((uint16_t)(w[i] >> 16))); }
    }
}
__device__ __forceinline__ void ld8h(const uint16_t* p, float (&v)[8], bool bf) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (bf) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
        else { v[2 * i] = __half2float(__ushort_as_half((uint16_t)w[i])); v[2 * i + 1] = __half2float(__ushort_as_half((uint16_t)(w[i] >> 16))); }
    }
}
__device__ __forceinline__ uint16_t cvt_h(float f, bool bf) {
    if (bf) { union { __hip_bfloat16 h; uint16_t u; } c; c.h = __float2bfloat16(f); return c.u; }
    return __half_as_ushort(__float2half_rn(f));
}

// x2 != null: the normalised tensor is the channel concatenation [x (Ca channels) | x2 (C - Ca channels)] read in place
__global__ __launch_bounds__(512) void gsw_gn_pf_stats_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int32_t Ca,
                                                             float* __restrict__ partial, int32_t C, int32_t G,
                                                             int32_t HpWp, int32_t slab_len, int32_t P, int bf) {
    // Deterministic reduction (no atomics: two runs on the same input give the same bits): every thread parks its 8 per-channel sums in LDS,
    // one thread per channel folds the P pixel lanes in a fixed order, one thread per group folds the group's channels in a fixed order.
    __shared__ float s_part[2][512 * 8];
    const int32_t b = blockIdx.y, s = blockIdx.x, nslab = gridDim.x;
    const int32_t cv = C >> 3, tid = threadIdx.x;
    const int32_t cvec = tid % cv, prow = tid / cv;
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int32_t i0 = s * slab_len, i1 = min(HpWp, i0 + slab_len);
    const bool second = x2 && cvec * 8 >= Ca;
    const int32_t ld = x2 ? (second ? C - Ca : Ca) : C;
    const uint16_t* base = (second ? x2 + (cvec * 8 - Ca) : x + cvec * 8) + ((int64_t)b * HpWp) * ld;
    int32_t i = i0 + prow;
    for (; i + 3 * P < i1; i += 4 * P) {          // four independent 16-byte loads in flight per thread (one alone reaches ~2 TB/s)
        uint4 u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const uint4*>(base + (int64_t)(i + j * P) * ld);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
            up8h(u[j], v, bf);
#pragma unroll
            for (int k = 0; k < 8; ++k) { sum[k] += v[k]; sq[k] = fmaf(v[k], v[k], sq[k]); }
        }
    }
    for (; i < i1; i += P) {
        float v[8];
        ld8h(base + (int64_t)i * ld, v, bf);
#pragma unroll
        for (int k = 0; k < 8; ++k) { sum[k] += v[k]; sq[k] = fmaf(v[k], v[k], sq[k]); }
    }
    {
        float4* ps = reinterpret_cast<float4*>(&s_part[0][(prow * cv + cvec) * 8]);
        float4* pq = reinterpret_cast<float4*>(&s_part[1][(prow * cv + cvec) * 8]);
        ps[0] = make_float4(sum[0], sum[1], sum[2], sum[3]); ps[1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
        pq[0] = make_float4(sq[0], sq[1], sq[2], sq[3]); pq[1] = make_float4(sq[4], sq[5], sq[6], sq[7]);
    }
    __syncthreads();
    const int32_t nthr = cv * P;
    for (int32_t c = tid; c < 2 * C; c += nthr) {           // channel c (sums), then channel c - C (squares): fold the pixel lanes into lane 0's slot
        float* col = &s_part[c >= C ? 1 : 0][c >= C ? c - C : c];
        float a = col[0];
        for (int32_t pr = 1; pr < P; ++pr) a += col[pr * C];
        col[0] = a;
    }
    __syncthreads();
    const int32_t cpg = C / G;
    if (tid < 2 * G) {
        const int32_t g = tid >= G ? tid - G : tid;
        const float* col = &s_part[tid >= G ? 1 : 0][g * cpg];
        float a = col[0];
        for (int32_t k = 1; k < cpg; ++k) a += col[k];
        partial[(((int64_t)b * nslab + s) * G + g) * 2 + (tid >= G ? 1 : 0)] = a;
    }
}

__global__ __launch_bounds__(512) void gsw_gn_pf_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ x2, int32_t Ca,
                                                             const float* __restrict__ partial, const uint16_t* __restrict__ gamma,
                                                             const uint16_t* __restrict__ beta, uint16_t* __restrict__ y, int32_t C, int32_t G, int32_t Hp, int32_t Wp,
                                                             int32_t nslab_stats, int32_t slab_len, int32_t P, float eps, int act, int tokens, int bf) {
    __shared__ float s_mean[GN_MAX_GROUPS], s_rstd[GN_MAX_GROUPS];
    const int32_t b = blockIdx.y, s = blockIdx.x;
    const int32_t cv = C >> 3, tid = threadIdx.x;
    const int32_t cvec = tid % cv, prow = tid / cv;
    const int32_t HpWp = Hp * Wp, H = Hp - 2, W = Wp - 2, cpg = C / G;
    // Fold the statistics of every group: L lanes per group add every L-th record (independent loads in flight -- one thread per group walking up to 64
    // slab records one after the other was most of this kernel's 18 us on a single image), then lane order: a fixed order, reproducible.
    __shared__ float s_fs[16][GN_MAX_GROUPS], s_fq[16][GN_MAX_GROUPS];
    int32_t L = 1;
    while (L < 16 && 2 * L * G <= (int32_t)blockDim.x) L <<= 1;
    if (tid < L * G) {
        const int32_t g = tid % G, j = tid / G;
        float sm = 0.f, sq = 0.f;
        if (nslab_stats > 0) {
            for (int32_t k = j; k < nslab_stats; k += L) {
                const float2 o = *reinterpret_cast<const float2*>(partial + (((int64_t)b * nslab_stats + k) * G + g) * 2);
                sm += o.x; sq += o.y;
            }
        } else {                                 // statistics as per-column-PAIR sums [B][C / 2][2] (gsw_gn_colstats_finish_kernel): this group's pairs
            const float2* o = reinterpret_cast<const float2*>(partial) + (int64_t)b * (C >> 1) + g * (cpg >> 1);
            for (int32_t k = j; k < (cpg >> 1); k += L) { sm += o[k].x; sq += o[k].y; }
        }
        s_fs[j][g] = sm; s_fq[j][g] = sq;
    }
    __syncthreads();
    if (tid < G) {
        float sm = s_fs[0][tid], sq = s_fq[0][tid];
        for (int32_t j = 1; j < L; ++j) { sm += s_fs[j][tid]; sq += s_fq[j][tid]; }
        const float n = (float)(H * W * cpg);
        const float mean = sm / n;
        const float var = fmaxf(sq / n - mean * mean, 0.f);
        s_mean[tid] = mean; s_rstd[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    float sc[8], sh[8];
    {
        float ga[8], be[8];
        ld8h(gamma + cvec * 8, ga, bf);
        ld8h(beta + cvec * 8, be, bf);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int32_t g = (cvec * 8 + k) / cpg;
            sc[k] = ga[k] * s_rstd[g];
            sh[k] = be[k] - s_mean[g] * sc[k];
        }
    }
    const int32_t i0 = s * slab_len, i1 = min(HpWp, i0 + slab_len);
    const bool second = x2 && cvec * 8 >= Ca;
    const int32_t ld = x2 ? (second ? C - Ca : Ca) : C;
    const uint16_t* base = (second ? x2 + (cvec * 8 - Ca) : x + cvec * 8) + ((int64_t)b * HpWp) * ld;
    // One row = 16 bytes per thread.  The kernel was VALU-bound, not HBM-bound: an IEEE division per element for the sigmoid, an integer division
    // per row for (y, x), one load in flight.  Now: sigmoid = v_rcp(1 + v_exp(-t log2 e)), (y, x) carried incrementally, four rows in flight.
    auto emit = [&](const uint4& u, int32_t i, int32_t yy, int32_t xx) {
        const bool border = (yy == 0) | (yy == Hp - 1) | (xx == 0) | (xx == Wp - 1);
        float v[8];
        up8h(u, v, bf);
        uint16_t h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = fmaf(v[k], sc[k], sh[k]);
            if (act) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
            h[k] = cvt_h(t, bf);
        }
        uint4 o = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                             (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
        if (border) o = make_uint4(0, 0, 0, 0);              // (border rows of a PF input are zeros in valid memory: loaded unconditionally)
        if (tokens) {
            if (!border) *reinterpret_cast<uint4*>(y + (((int64_t)b * H + (yy - 1)) * W + (xx - 1)) * C + cvec * 8) = o;
        } else {
            *reinterpret_cast<uint4*>(y + ((int64_t)b * HpWp + i) * C + cvec * 8) = o;
        }
    };
    int32_t i = i0 + prow;
    int32_t yy = i / Wp, xx = i - yy * Wp;
    auto advance = [&]() { xx += P; while (xx >= Wp) { xx -= Wp; ++yy; } };
    for (; i + 3 * P < i1; i += 4 * P) {
        uint4 u[4];
        int32_t ys[4], xs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u[j] = *reinterpret_cast<const uint4*>(base + (int64_t)(i + j * P) * ld);
            ys[j] = yy; xs[j] = xx;
            advance();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) emit(u[j], i + j * P, ys[j], xs[j]);
    }
    for (; i < i1; i += P) {
        const uint4 u = *reinterpret_cast<const uint4*>(base + (int64_t)i * ld);
        emit(u, i, yy, xx);
        advance();
    }
}

// ------------------------------------------------------------------------------------------------
// Residual add + LayerNorm of the transformer blocks: x_new = x + delta (optional), y = LN(x_new) * gamma + beta.
// L lanes per token row (L = 8 / 16 / 32 / 64 for C <= 320 / 640 / 1280 / 1536), i.e. 64/L rows per wave and up to NVL = 5 16-byte
// vectors per lane: every lane is busy at the UNet's widths and a wave keeps 10 loads per lane in flight (one row per wave kept a
// single 640-byte row in flight and ran at 2.7 TB/s).  The row lives in registers, mean and centred variance are exact two-pass fp32
// with log2(L) exchange steps each, gamma / beta stay packed in registers across the rows a wave processes.  4 HBM passes (read x,
// delta; write x_new, y) instead of add (3) + LayerNorm (2).
// ------------------------------------------------------------------------------------------------
constexpr int LN_NVL = 5;
template <int L>
__global__ __launch_bounds__(256) void gsw_add_layernorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ delta, const uint16_t* __restrict__ gamma,
                                                               const uint16_t* __restrict__ beta, uint16_t* __restrict__ xnew, uint16_t* __restrict__ y, int64_t rows,
                                                               int32_t C, float eps, int bf) {
    constexpr int RPW = 64 / L;                                  // rows per wave
    const int32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t sub = lane & (L - 1), rsel = lane / L;
    const int32_t nv = C >> 3;
    uint4 gpk[LN_NVL], bpk[LN_NVL];
#pragma unroll
    for (int it = 0; it < LN_NVL; ++it) {
        const int32_t v = sub + L * it;
        gpk[it] = v < nv ? *reinterpret_cast<const uint4*>(gamma + v * 8) : make_uint4(0, 0, 0, 0);
        bpk[it] = v < nv ? *reinterpret_cast<const uint4*>(beta + v * 8) : make_uint4(0, 0, 0, 0);
    }
    const float invC = 1.0f / (float)C;
    const int64_t rstep = (int64_t)gridDim.x * 4 * RPW;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += rstep) {
        const int64_t row = row0 + rsel;
        const bool live = row < rows;                             // the exchange steps below need every lane of the wave
        float val[LN_NVL][8];
        uint4 xin[LN_NVL], din[LN_NVL];
#pragma unroll
        for (int it = 0; it < LN_NVL; ++it) {                     // all loads first: 2 * NVL independent 16-byte loads in flight per lane
            const int32_t v = sub + L * it;
            const bool ok = live && v < nv;
            xin[it] = ok ? *reinterpret_cast<const uint4*>(x + row * C + v * 8) : make_uint4(0, 0, 0, 0);
            din[it] = (ok && delta) ? *reinterpret_cast<const uint4*>(delta + row * C + v * 8) : make_uint4(0, 0, 0, 0);
        }
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < LN_NVL; ++it) {
            const int32_t v = sub + L * it;
            up8h(xin[it], val[it], bf);
            if (delta) {
                float d[8];
                up8h(din[it], d, bf);
                uint16_t h[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) h[k] = cvt_h(val[it][k] + d[k], bf);
                // LN sees the stored (rounded) sum, exactly like add followed by LayerNorm
                const uint4 o = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                           (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
                if (live && v < nv) *reinterpret_cast<uint4*>(xnew + row * C + v * 8) = o;
                up8h(o, val[it], bf);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += val[it][k];        // vectors past nv are zeros
        }
#pragma unroll
        for (int s_ = L / 2; s_ > 0; s_ >>= 1) sum += __shfl_xor(sum, s_, 64);
        const float mean = sum * invC;
        float sq = 0.f;
#pragma unroll
        for (int it = 0; it < LN_NVL; ++it) {
            if (sub + L * it < nv) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = val[it][k] - mean; sq = fmaf(d, d, sq); }
            }
        }
#pragma unroll
        for (int s_ = L / 2; s_ > 0; s_ >>= 1) sq += __shfl_xor(sq, s_, 64);
        const float rstd = rsqrtf(sq * invC + eps);
#pragma unroll
        for (int it = 0; it < LN_NVL; ++it) {
            const int32_t v = sub + L * it;
            if (live && v < nv) {
                const uint32_t gw[4] = {gpk[it].x, gpk[it].y, gpk[it].z, gpk[it].w}, bw[4] = {bpk[it].x, bpk[it].y, bpk[it].z, bpk[it].w};
                uint16_t h[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float g0, g1, b0, b1;
                    if (bf) { g0 = __uint_as_float(gw[i] << 16); g1 = __uint_as_float(gw[i] & 0xFFFF0000u); b0 = __uint_as_float(bw[i] << 16); b1 = __uint_as_float(bw[i] & 0xFFFF0000u); }
                    else { g0 = __half2float(__ushort_as_half((uint16_t)gw[i])); g1 = __half2float(__ushort_as_half((uint16_t)(gw[i] >> 16)));
                           b0 = __half2float(__ushort_as_half((uint16_t)bw[i])); b1 = __half2float(__ushort_as_half((uint16_t)(bw[i] >> 16))); }
                    h[2 * i] = cvt_h(fmaf((val[it][2 * i] - mean) * rstd, g0, b0), bf);
                    h[2 * i + 1] = cvt_h(fmaf((val[it][2 * i + 1] - mean) * rstd, g1, b1), bf);
                }
                *reinterpret_cast<uint4*>(y + row * C + v * 8) = make_uint4((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16),
                                                                            (uint32_t)h[4] | ((uint32_t)h[5] << 16), (uint32_t)h[6] | ((uint32_t)h[7] << 16));
            }
        }
    }
}

// host ---------------------------------------------------------------------------------------------
extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip; read by gsw_last_hip_error()
#define g_conv_hip_error g_last_hip_error
#define GSW_CONV_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { g_conv_hip_error = (int)_e; return GSW_ERR_HIP; } } while (0)
static int launch_engine(const ConvArgs& a, int64_t M, int N, int dtype, void* stream, GswMmExtras* ex);
static bool use_engine(const ConvArgs& a, int N);

// Convolutions on the matmul engine (csrc/gswm_mm.hip): every tap is a row offset into the padded-flat activation, so the 3x3 (or
// the 2x2 sub-pixel) convolution is a GEMM whose K dimension walks (channel block, tap row, tap) -- no halo tile, the activation slab of
// each tap comes through L2.  (The round-1 / round-2 halo and 128 x 160 kernels this replaced are gone: profiles/r02h_conv_engine_vs_halo.txt
// holds the last A/B.)
// Zero the border rows of a padded-flat tensor [B, Hp, Wp, N] (row stride N): top and bottom rows, first and last column.
__global__ __launch_bounds__(256) void gsw_pf_zero_border_kernel(uint16_t* __restrict__ y, int B, int Hp, int Wp, int n8) {
    const int nb = 2 * Wp + 2 * (Hp - 2);
    const int64_t total = (int64_t)B * nb * n8;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int c8 = (int)(u % n8);
        const int64_t t = u / n8;
        const int j = (int)(t % nb), b = (int)(t / nb);
        int row;
        if (j < Wp) row = j;
        else if (j < 2 * Wp) row = (Hp - 1) * Wp + (j - Wp);
        else { const int k = j - 2 * Wp; row = (1 + (k >> 1)) * Wp + ((k & 1) ? Wp - 1 : 0); }
        reinterpret_cast<uint4*>(y + ((int64_t)b * Hp * Wp + row) * (int64_t)n8 * 8)[c8] = make_uint4(0, 0, 0, 0);
    }
}

static void zero_border(void* y, int B, int Hp, int Wp, int N, hipStream_t st) {
    const int64_t total = (int64_t)B * (2 * Wp + 2 * (Hp - 2)) * (N / 8);
    const uint32_t grid = (uint32_t)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(gsw_pf_zero_border_kernel, dim3(grid), dim3(256), 0, st, (uint16_t*)y, B, Hp, Wp, N / 8);
}

static int launch_engine(const ConvArgs& a, int64_t M, int N, int dtype, void* stream, GswMmExtras* ex) {
    MMArgs m;
    const int tw = a.ntaps == 9 ? 3 : a.ntaps == 4 ? 2 : 1;
    m.seg[0] = MMSeg{a.x, a.ldx, a.C / 64, a.ntaps, tw, a.in_Wp, a.tap_off[0], 0};
    m.seg[1] = MMSeg{a.x1 ? a.x1 : a.x, a.C1 ? a.C1 : a.ldx, a.C1 / 64, 1, 1, 0, 0, a.ntaps * a.C};
    m.seg[2] = MMSeg{a.x2 ? a.x2 : a.x, a.C2 ? a.C2 : a.ldx, a.C2 / 64, 1, 1, 0, 0, a.ntaps * a.C + a.C1};
    m.nseg = 1 + (a.C1 > 0) + (a.C2 > 0);
    m.P = a.ntaps * (a.C / 64) + a.C1 / 64 + a.C2 / 64;
    m.w = a.w; m.ldw = a.ntaps * a.C + a.C1 + a.C2;
    // the M dimension enumerates interior pixels only (the padded border is 6 % of the rows at 64x64 and 56 % at 8x8); the border rows of
    // the output are zeroed by a separate small kernel (the up2x caller does that once for its four parity launches).
    // Small tensors (one or two images: the launch is latency-bound, not throughput-bound) enumerate ALL padded rows instead and let the epilogue
    // write the zeros of the border rows itself: one kernel less per convolution (62 per forward; a launch costs ~6 us of a 6.6 ms one-image forward).
    // Border rows read their taps from neighbouring / guard rows whose contents are arbitrary -- their accumulators are discarded.
    // In between (8-64 images at the deep levels) the choice follows the engine's own plan: the interior enumeration has fewer row tiles, which can be what lets the
    // launch split K over the whole chip (16 images at 16 x 16: 5184 padded rows are 168 tiles of 256, 4096 interior rows 128 -> 2 x 128 workgroups), for one
    // border kernel (~5 us) more.
    const int B = (int)(M / ((int64_t)a.Hp * a.Wp));
    const int64_t M_int = (int64_t)B * (a.Hp - 2) * (a.Wp - 2);
    bool whole = !a.up && a.stride == 1 && M <= 8192;
    if (whole && M > 1024 && !(ex && ex->max_splits > 1))
        whole = gsw_mm_predict_us(M, N, m.P, ex) <= gsw_mm_predict_us(M_int, N, m.P, ex) + 5.0;
    m.M = whole ? (int32_t)M : (int32_t)M_int; m.N = N;
    m.flags = whole ? MM_FLAG_NONE : MM_FLAG_COMPACT;
    m.bias = a.bias; m.rowbias = a.rowbias; m.resid = a.resid; m.y = a.y; m.colstats = nullptr; m.y2 = nullptr; m.n_rows = 0; m.ln_stat = nullptr; m.ln_u = nullptr; m.ln_v = nullptr;
    m.ldy = N; m.ldr = N; m.ldrb = a.ldrb;
    m.mode = a.up ? MM_MODE_UP2X : MM_MODE_PF;
    m.Hp = a.Hp; m.Wp = a.Wp; m.in_Hp = a.in_Hp; m.in_Wp = a.in_Wp; m.stride = a.stride; m.S = 1; m.Wimg = 1; m.up = a.up;
    const int rc = gsw_mm_launch(m, dtype, stream, ex);
    // the border of the output: zeroed by a small kernel of its own (it touches rows the engine launch does not) -- unless the caller declared the output
    // GroupNorm-only (GSW_MM_GN_ONLY) AND this launch wrote the column records that GroupNorm will take its statistics from: nothing reads the border then
    const bool gn_only = ex && (ex->flags & GSW_MM_GN_ONLY) && ex->colstats_rows_per_block > 0;
    if (rc == GSW_OK && !a.up && !whole && !gn_only) zero_border(a.y, B, a.Hp, a.Wp, N, (hipStream_t)stream);
    return rc;
}

static bool use_engine(const ConvArgs& a, int N) {
    // N: any multiple of 8 from 128 up (a partial last 160-column tile costs a full one: 128 / 256 / 512 channels of the VAE run at 80 %);
    // narrower outputs (the 4-channel edge padded to 64) stay on the 64-column kernel
    if (a.stride == 2 && (a.ntaps != 9 || a.C1 || a.C2 || a.up)) return false;
    return (a.ntaps == 9 || a.ntaps == 4 || a.ntaps == 1) && N % 8 == 0 && N >= 128 && a.C % 64 == 0 && a.C1 % 64 == 0 && a.C2 % 64 == 0;
}

static int launch_conv_gemm(ConvArgs& a, int64_t M, int N, int dtype, void* stream, GswMmExtras* ex);

int gsw_conv_pf(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                int B, int H, int W, int C, int N, int ksize, int stride, int ldx, int dtype, void* stream) {
    return gsw_conv_pf_ex(x_dev, w_dev, bias_dev, rowbias_dev, ld_rowbias, resid_dev, y_dev, B, H, W, C, N, ksize, stride, ldx, dtype, nullptr, stream);
}

int gsw_conv_pf_ex(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                   int B, int H, int W, int C, int N, int ksize, int stride, int ldx, int dtype, GswMmExtras* ex, void* stream) {
    // H, W: OUTPUT spatial size; input spatial size is (H*stride, W*stride)
    if (!x_dev || !w_dev || !y_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (ksize == 1 && stride != 1)) return GSW_ERR_BAD_ARG;
    if (C % CV_BK || N % 8 || ldx < C || (ldx & 7)) return GSW_ERR_UNSUPPORTED;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (rowbias_dev && ld_rowbias != 0 && (ld_rowbias < N || (ld_rowbias & 7))) return GSW_ERR_BAD_ARG;
    ConvArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.rowbias = rowbias_dev; a.resid = resid_dev; a.y = y_dev;
    a.ldrb = rowbias_dev && ld_rowbias ? ld_rowbias : N;
    a.C = C; a.N = N; a.Hp = H + 2; a.Wp = W + 2; a.stride = stride; a.ldx = ldx;
    a.in_Hp = H * stride + 2; a.in_Wp = W * stride + 2;
    const int64_t M = (int64_t)B * a.Hp * a.Wp;
    if (M > 0x7FFFFF00 || ((int64_t)B * a.in_Hp * a.in_Wp + 2 * a.in_Wp) * ldx >= ((int64_t)1 << 31) || (int64_t)N * ksize * ksize * C >= ((int64_t)1 << 31))
        return GSW_ERR_UNSUPPORTED;     // 32-bit element offsets inside the kernel
    a.M = (int32_t)M;
    a.ntaps = ksize * ksize;
    for (int i = 0; i < 9; ++i) a.tap_off[i] = 0;
    if (ksize == 3) {
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw)
                a.tap_off[kh * 3 + kw] = stride == 1 ? (kh - 1) * a.in_Wp + (kw - 1) : kh * a.in_Wp + kw;
    }
    a.dense = 0; a.geglu = 0; a.ldy = N;
    a.x1 = nullptr; a.x2 = nullptr; a.C1 = 0; a.C2 = 0; a.up = 0;
    if (N % CV_BN && !use_engine(a, N)) return GSW_ERR_UNSUPPORTED;      // the round-1 kernels tile N by 64; the engine takes any N % 8 from 128 up
    return launch_conv_gemm(a, M, N, dtype, stream, ex);
}

static int launch_conv_gemm(ConvArgs& a, int64_t M, int N, int dtype, void* stream, GswMmExtras* ex) {
    hipStream_t st = (hipStream_t)stream;
    if (use_engine(a, N)) return launch_engine(a, M, N, dtype, stream, ex);
    // off the engine: no records, no split-K
    if (ex) { ex->colstats_rows_per_block = 0; ex->colstats_blocks = 0; ex->rowstats_slots = 0; ex->splits = 1; }
    if (N % CV_BN || a.C1 || a.C2 || a.up) return GSW_ERR_UNSUPPORTED;
    const uint32_t grid = (uint32_t)(((M + CV_BM - 1) / CV_BM) * (N / CV_BN));
    if (dtype == GSW_F16) hipLaunchKernelGGL((gsw_conv_gemm_kernel<_Float16>), dim3(grid), dim3(CV_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gsw_conv_gemm_kernel<__bf16>), dim3(grid), dim3(CV_THREADS), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_conv_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

// GroupNorm statistics from the column records the producing engine launches left behind (MMArgs::colstats): per image, fold the image's row
// blocks (and the four parity launches of a sub-pixel upsampler) per column in a fixed order, then the columns of every group -> the
// [B][1][groups][2] (sum, sum of squares) record the apply kernel reads.  Two sources = the channel concatenation [x | x2].
// Work split of the apply kernel: slabs of >= 4 P rows (one iteration of its four-rows-in-flight loop), ~2048 workgroups in total at most.  With one image
// the 64-slab split of the statistics pass left 3/4 of the chip idle and every workgroup two dependent iterations long (18.8 us per launch at 64 x 64).
static void gn_apply_slabs(int B, int HpWp, int P, int& nslab, int& slab_len) {
    nslab = std::max(1, std::min((2048 + B - 1) / B, (HpWp + 4 * P - 1) / (4 * P)));
    slab_len = (HpWp + nslab - 1) / nslab;
    nslab = (HpWp + slab_len - 1) / slab_len;
}

struct GnColSrc { const float* cs; int32_t C, npar, bpi, nblk; };      // records [npar][nblk][2 planes: sums | sums of squares][C / 2 column pairs]
// grid (B, ceil(C / 2 / 64)), 512 threads = 64 column pairs x 8 block lanes: lane j adds blocks j, j + 8, ... (independent loads in flight), the eight
// lanes are folded in a fixed order -> pairsum[b][cp] = (sum, sum of squares) of columns 2 cp, 2 cp + 1 over image b.  The apply kernel folds the
// pairs of a group (groups are an even number of channels wide).
__global__ __launch_bounds__(512) void gsw_gn_colstats_finish_kernel(GnColSrc s1, GnColSrc s2, float2* __restrict__ pairsum, int32_t C) {
    __shared__ float2 part[8][64];
    const int32_t b = blockIdx.x, tid = threadIdx.x, cl = tid & 63, lane = tid >> 6;
    const int32_t cp = blockIdx.y * 64 + cl, np = C >> 1;
    float a = 0.f, q = 0.f;
    if (cp < np) {
        const bool first = cp < (s1.C >> 1);
        const GnColSrc& s = first ? s1 : s2;
        const int32_t cc = first ? cp : cp - (s1.C >> 1), h = s.C >> 1;
        for (int32_t par = 0; par < s.npar; ++par) {
            const float* rec = s.cs + ((int64_t)par * s.nblk + (int64_t)b * s.bpi) * s.C + cc;      // block stride: 2 planes x C / 2 = C floats
#pragma unroll 4
            for (int32_t k = lane; k < s.bpi; k += 8) { a += rec[(int64_t)k * s.C]; q += rec[(int64_t)k * s.C + h]; }
        }
    }
    part[lane][cl] = make_float2(a, q);
    __syncthreads();
    if (lane == 0 && cp < np) {
        float2 t = part[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) { t.x += part[j][cl].x; t.y += part[j][cl].y; }
        pairsum[(int64_t)b * np + cp] = t;
    }
}

int gsw_groupnorm_pf_cs(const void* x_dev, const void* x2_dev, int Ca, const float* cs1_dev, int cs1_rows, int cs1_npar, int cs1_blocks,
                        const float* cs2_dev, int cs2_rows, int cs2_npar, int cs2_blocks, const void* gamma_dev, const void* beta_dev, void* out_dev,
                        float* workspace_dev, int B, int H, int W, int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream) {
    // like gsw_groupnorm_pf2, with the statistics taken from column records instead of a pass over the tensor(s).
    // csN_rows: rows per record block (32 / 64), csN_npar: 1, or 4 for the output of gsw_conv_up2x_pf (records per parity launch over the
    // LOW-resolution pixels), csN_blocks: blocks per parity buffer.  H * W (per parity: H/2 * W/2) must be a multiple of csN_rows.
    if (x2_dev && (Ca <= 0 || Ca >= C || (Ca & 7) || !cs2_dev)) return GSW_ERR_BAD_ARG;
    if (!x_dev || !cs1_dev || !gamma_dev || !beta_dev || !out_dev || !workspace_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0) return GSW_ERR_BAD_ARG;
    if ((C & 7) || C % groups || groups > GN_MAX_GROUPS || (C >> 3) > 512 || C > 4096 || ((C / groups) & 1)) return GSW_ERR_UNSUPPORTED;      // column pairs: even group width
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    const int C1 = x2_dev ? Ca : C;
    auto src = [&](const float* cs, int Cs, int rows, int npar, int blocks, GnColSrc& o) -> int {
        if (rows <= 0 || (npar != 1 && npar != 4) || blocks <= 0) return GSW_ERR_BAD_ARG;
        const int64_t pix = npar == 4 ? ((H & 1) || (W & 1) ? -1 : (int64_t)(H / 2) * (W / 2)) : (int64_t)H * W;
        if (pix <= 0 || pix % rows || (int64_t)B * (pix / rows) > blocks) return GSW_ERR_UNSUPPORTED;
        o = GnColSrc{cs, Cs, npar, (int32_t)(pix / rows), blocks};
        return GSW_OK;
    };
    GnColSrc s1, s2 = GnColSrc{nullptr, 0, 0, 0, 0};
    { const int rc = src(cs1_dev, C1, cs1_rows, cs1_npar, cs1_blocks, s1); if (rc != GSW_OK) return rc; }
    if (x2_dev) { const int rc = src(cs2_dev, C - Ca, cs2_rows, cs2_npar, cs2_blocks, s2); if (rc != GSW_OK) return rc; }
    const int cv = C >> 3;
    const int P = std::max(1, 320 / cv);
    const int threads = cv * P;
    const int HpWp = (H + 2) * (W + 2);
    int nslab, slab_len;
    gn_apply_slabs(B, HpWp, P, nslab, slab_len);
    hipStream_t st = (hipStream_t)stream;
    const int bf = dtype == GSW_BF16;
    // workspace_dev: >= B * C floats here (the column-pair sums)
    hipLaunchKernelGGL(gsw_gn_colstats_finish_kernel, dim3(B, (C / 2 + 63) / 64), dim3(512), 0, st, s1, s2, reinterpret_cast<float2*>(workspace_dev), C);
    hipLaunchKernelGGL(gsw_gn_pf_apply_kernel, dim3(nslab, B), dim3(threads), 0, st, (const uint16_t*)x_dev, (const uint16_t*)x2_dev, Ca, (const float*)workspace_dev, (const uint16_t*)gamma_dev,
                       (const uint16_t*)beta_dev, (uint16_t*)out_dev, C, groups, H + 2, W + 2, 0, slab_len, P, eps, act, out_tokens, bf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_conv_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_gn_colstats_pairs(const float* cs_dev, int cs_rows, int cs_npar, int cs_blocks, float* pairsum_dev, int B, int H, int W, int C, void* stream) {
    // the first half of gsw_groupnorm_pf_cs on its own: column records of the producing launch -> per-image, per-column-PAIR (sum, sum of squares) [B][C / 2][2]
    // (fixed fold order); the consumer is a kernel that applies the GroupNorm itself (gsw_gn_proj_tokens)
    if (!cs_dev || !pairsum_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || cs_rows <= 0 || (cs_npar != 1 && cs_npar != 4) || cs_blocks <= 0) return GSW_ERR_BAD_ARG;
    if ((C & 7) || C > 4096) return GSW_ERR_UNSUPPORTED;
    const int64_t pix = cs_npar == 4 ? ((H & 1) || (W & 1) ? -1 : (int64_t)(H / 2) * (W / 2)) : (int64_t)H * W;
    if (pix <= 0 || pix % cs_rows || (int64_t)B * (pix / cs_rows) > cs_blocks) return GSW_ERR_UNSUPPORTED;
    const GnColSrc s1 = GnColSrc{cs_dev, C, cs_npar, (int32_t)(pix / cs_rows), cs_blocks}, s2 = GnColSrc{nullptr, 0, 0, 0, 0};
    hipLaunchKernelGGL(gsw_gn_colstats_finish_kernel, dim3(B, (C / 2 + 63) / 64), dim3(512), 0, (hipStream_t)stream, s1, s2, reinterpret_cast<float2*>(pairsum_dev), C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_conv_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_groupnorm_pf2(const void* x_dev, const void* x2_dev, int Ca, const void* gamma_dev, const void* beta_dev, void* out_dev, float* workspace_dev,
                      int B, int H, int W, int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream) {
    // x2_dev != NULL: GroupNorm over the channel concatenation [x (Ca) | x2 (C - Ca)] without materialising it
    if (x2_dev && (Ca <= 0 || Ca >= C || (Ca & 7))) return GSW_ERR_BAD_ARG;
    // workspace_dev: >= B * 64 * groups * 2 floats
    if (!x_dev || !gamma_dev || !beta_dev || !out_dev || !workspace_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0) return GSW_ERR_BAD_ARG;
    if ((C & 7) || C % groups || groups > GN_MAX_GROUPS || (C >> 3) > 512) return GSW_ERR_UNSUPPORTED;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    const int cv = C >> 3;
    const int P = std::max(1, 320 / cv);
    const int threads = cv * P;
    const int HpWp = (H + 2) * (W + 2);
    // statistics pass: enough workgroups to fill the chip (~2048 in total), at most 64 slabs per image (workspace bound)
    int nslab = std::max(1, std::min(64, std::min((2048 + B - 1) / B, (HpWp + P - 1) / P)));
    const int slab_len = (HpWp + nslab - 1) / nslab;
    nslab = (HpWp + slab_len - 1) / slab_len;
    int nslab_a, slab_len_a;                 // the apply pass splits the image on its own (one image: up to ~140 workgroups of one iteration each)
    gn_apply_slabs(B, HpWp, P, nslab_a, slab_len_a);
    hipStream_t st = (hipStream_t)stream;
    const int bf = dtype == GSW_BF16;
    hipLaunchKernelGGL(gsw_gn_pf_stats_kernel, dim3(nslab, B), dim3(threads), 0, st, (const uint16_t*)x_dev, (const uint16_t*)x2_dev, Ca, workspace_dev, C, groups, HpWp, slab_len, P, bf);
    hipLaunchKernelGGL(gsw_gn_pf_apply_kernel, dim3(nslab_a, B), dim3(threads), 0, st, (const uint16_t*)x_dev, (const uint16_t*)x2_dev, Ca, (const float*)workspace_dev, (const uint16_t*)gamma_dev,
                       (const uint16_t*)beta_dev, (uint16_t*)out_dev, C, groups, H + 2, W + 2, nslab, slab_len_a, P, eps, act, out_tokens, bf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_conv_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_groupnorm_pf(const void* x_dev, const void* gamma_dev, const void* beta_dev, void* out_dev, float* workspace_dev, int B, int H, int W,
                     int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream) {
    return gsw_groupnorm_pf2(x_dev, nullptr, 0, gamma_dev, beta_dev, out_dev, workspace_dev, B, H, W, C, groups, eps, act, out_tokens, dtype, stream);
}

int gsw_add_layernorm(const void* x_dev, const void* delta_dev, const void* gamma_dev, const void* beta_dev, void* xnew_dev, void* y_dev,
                      int64_t rows, int C, float eps, int dtype, void* stream) {
    if (!x_dev || !gamma_dev || !beta_dev || !y_dev || rows < 0 || C <= 0 || (delta_dev && !xnew_dev)) return GSW_ERR_BAD_ARG;
    if ((C & 7) || C > 1536) return GSW_ERR_UNSUPPORTED;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (rows == 0) return GSW_OK;
    const int nv = C >> 3;
    const int L = nv <= 8 * LN_NVL ? 8 : nv <= 16 * LN_NVL ? 16 : nv <= 32 * LN_NVL ? 32 : 64;       // lanes per row; C <= 1536 -> nv <= 192 <= 64 * 5
    const int64_t rows_per_block = 4 * (64 / L);
    const uint32_t grid = (uint32_t)std::min<int64_t>((rows + rows_per_block - 1) / rows_per_block, 256 * 16);
#define GSW_LN_LAUNCH(LL)                                                                                                                  \
    hipLaunchKernelGGL(gsw_add_layernorm_kernel<LL>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_dev, (const uint16_t*)delta_dev, \
                       (const uint16_t*)gamma_dev, (const uint16_t*)beta_dev, (uint16_t*)xnew_dev, (uint16_t*)y_dev, rows, C, eps, dtype == GSW_BF16)
    if (L == 8) GSW_LN_LAUNCH(8); else if (L == 16) GSW_LN_LAUNCH(16); else if (L == 32) GSW_LN_LAUNCH(32); else GSW_LN_LAUNCH(64);
#undef GSW_LN_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_conv_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_conv3x3_res_pf(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                       int B, int H, int W, int C, int N, const void* x1_dev, int C1, const void* x2_dev, int C2, int dtype, void* stream) {
    return gsw_conv3x3_res_pf_ex(x_dev, w_dev, bias_dev, rowbias_dev, ld_rowbias, resid_dev, y_dev, B, H, W, C, N, x1_dev, C1, x2_dev, C2, dtype, nullptr, stream);
}

int gsw_conv3x3_res_pf_ex(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                          int B, int H, int W, int C, int N, const void* x1_dev, int C1, const void* x2_dev, int C2, int dtype, GswMmExtras* ex, void* stream) {
    // 3x3 stride-1 convolution of x plus 1x1 convolutions of x1 (C1 channels) and x2 (C2 channels) in ONE GEMM:
    // w_dev = [N][9*C + C1 + C2].  The resnet's conv2 + conv_shortcut(cat(x1, x2)) + residual in a single kernel.
    if (!x_dev || !w_dev || !y_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if ((x1_dev && C1 <= 0) || (x2_dev && (C2 <= 0 || !x1_dev))) return GSW_ERR_BAD_ARG;
    if (C % CV_BK || N % 8 || N < 128 || (x1_dev && C1 % CV_BK) || (x2_dev && C2 % CV_BK)) return GSW_ERR_UNSUPPORTED;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (rowbias_dev && ld_rowbias != 0 && (ld_rowbias < N || (ld_rowbias & 7))) return GSW_ERR_BAD_ARG;
    ConvArgs a;
    a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.rowbias = rowbias_dev; a.resid = resid_dev; a.y = y_dev;
    a.ldrb = rowbias_dev && ld_rowbias ? ld_rowbias : N;
    a.C = C; a.N = N; a.Hp = H + 2; a.Wp = W + 2; a.stride = 1; a.ldx = C; a.in_Hp = a.Hp; a.in_Wp = a.Wp;
    const int64_t M = (int64_t)B * a.Hp * a.Wp;
    const int64_t cmax = std::max<int64_t>(C, std::max(C1, C2));
    if (M > 0x7FFFFF00 || (M + 2 * a.Wp) * cmax >= ((int64_t)1 << 31) || (int64_t)N * (9 * (int64_t)C + C1 + C2) >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    a.M = (int32_t)M; a.ntaps = 9;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) a.tap_off[kh * 3 + kw] = (kh - 1) * a.Wp + (kw - 1);
    a.dense = 0; a.geglu = 0; a.ldy = N;
    a.x1 = x1_dev; a.x2 = x2_dev; a.C1 = x1_dev ? C1 : 0; a.C2 = x2_dev ? C2 : 0; a.up = 0;
    return launch_engine(a, M, N, dtype, stream, ex);
}

int gsw_conv_up2x_pf(const void* x_dev, const void* w4_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int N, int dtype, void* stream) {
    return gsw_conv_up2x_pf_ex(x_dev, w4_dev, bias_dev, y_dev, B, H, W, C, N, dtype, nullptr, stream);
}

int gsw_conv_up2x_pf_ex(const void* x_dev, const void* w4_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int N, int dtype, GswMmExtras* ex_user,
                        void* stream) {
    // nearest-neighbour 2x upsampling followed by a 3x3 convolution (diffusers Upsample2D), computed from the LOW-resolution input:
    // output pixel (2i+dy, 2j+dx) only ever sees the 2x2 low-resolution neighbourhood (i+dy-1 .. i+dy, j+dx-1 .. j+dx), with the 3x3
    // weights summed over the taps that land on the same source pixel.  Four launches of the matmul engine (ntaps = 4, K = 4C), 2.25x
    // fewer FLOPs than convolving the upsampled tensor, and the upsampled tensor never exists.
    //   x: PF [B, H, W, C];  w4: [4 (dy*2+dx)][N][4 (a*2+b)][C] pre-summed weights;  y: PF [B, 2H, 2W, N] (border rows zeroed here).
    if (!x_dev || !w4_dev || !y_dev || B <= 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if (C % CV_BK || N % 8 || N < 128) return GSW_ERR_UNSUPPORTED;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    ConvArgs a;
    a.x = x_dev; a.bias = bias_dev; a.rowbias = nullptr; a.ldrb = N; a.resid = nullptr; a.y = y_dev;
    a.C = C; a.N = N; a.Hp = H + 2; a.Wp = W + 2; a.stride = 1; a.ldx = C; a.in_Hp = a.Hp; a.in_Wp = a.Wp;
    const int64_t M = (int64_t)B * a.Hp * a.Wp;
    const int64_t Mo = (int64_t)B * (2 * H + 2) * (2 * W + 2);
    if (M > 0x7FFFFF00 || (M + 2 * a.Wp) * C >= ((int64_t)1 << 31) || (int64_t)N * 4 * C >= ((int64_t)1 << 31) || Mo * N >= ((int64_t)1 << 40)) return GSW_ERR_UNSUPPORTED;
    a.M = (int32_t)M; a.ntaps = 4;
    a.dense = 0; a.geglu = 0; a.ldy = N;
    a.x1 = nullptr; a.x2 = nullptr; a.C1 = 0; a.C2 = 0;
    a.stride = 1; a.dense = 0; a.up = 1;
    zero_border(y_dev, B, 2 * H + 2, 2 * W + 2, N, (hipStream_t)stream);
    const size_t esz = 2;
    // a column-statistics request covers the whole output: each parity launch fills its quarter of the buffer ([4][blocks][N][2]); the four launches
    // report the same geometry (the last one's is handed back), or none of them writes records
    GswMmExtras none, *ex = ex_user;
    if (!ex) { gsw_mm_no_extras(&none); ex = &none; }
    float* const cs_base = ex->colstats_capacity > 0 ? ex->colstats_dev : nullptr;
    const int64_t cs_cap = cs_base ? ex->colstats_capacity : 0;
    GswMmExtras sub = *ex;
    struct Done { GswMmExtras* ex; GswMmExtras* sub; ~Done() {
        ex->colstats_rows_per_block = sub->colstats_rows_per_block; ex->colstats_blocks = sub->colstats_blocks; ex->rowstats_slots = 0; ex->splits = sub->splits; } } done{ex, &sub};
    sub.colstats_rows_per_block = 0; sub.colstats_blocks = 0; sub.splits = 1;
    for (int par = 0; par < 4; ++par) {
        sub.colstats_dev = cs_base ? cs_base + (size_t)par * (size_t)(cs_cap / 4) : nullptr;
        sub.colstats_capacity = cs_base ? cs_cap / 4 : 0;
        sub.rowstats_dev = nullptr; sub.rowstats_capacity = 0;
        const int dy = par >> 1, dx = par & 1;
        for (int i = 0; i < 9; ++i) a.tap_off[i] = 0;
        for (int ta = 0; ta < 2; ++ta)
            for (int tb = 0; tb < 2; ++tb) a.tap_off[ta * 2 + tb] = (ta + dy - 1) * a.Wp + (tb + dx - 1);
        a.w = (const uint8_t*)w4_dev + (size_t)par * N * 4 * C * esz;
        a.up = 1 + par;
        { const int rc = launch_engine(a, M, N, dtype, stream, &sub); if (rc != GSW_OK) return rc; }
    }
    return GSW_OK;
}
