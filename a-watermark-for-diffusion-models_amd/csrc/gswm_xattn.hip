// gswm_xattn.hip -- the cross-attention SUBLAYER of a transformer block as ONE launch (rows X2 / G1 of SURVEY.md section 8a):
//
//     x' = x + to_out( softmax( to_q(LayerNorm(x)) K^T / sqrt(d) ) V ) + b_out            K, V = to_k(ctx), to_v(ctx), 77 context tokens
//
// Reference call site: extract.py:66-69 and the generation loop (modified_stable_diffusion_gs.pyc) run diffusers' UNet2DConditionModel, whose
// BasicTransformerBlock.attn2 is this expression at every latent level.  Until round 5 it was three launches here -- the LayerNorm-folded query
// projection, the flash-attention kernel against 77 keys padded to 128 (0.07 of the MFMA peak: it streams q and o through HBM around a softmax whose
// K / V fit in 32 KiB) and the output projection + residual -- i.e. q and o, M x C each, written and re-read: 2.3 GB of traffic per launch trio at
// 64 x 64 for 0.67 GB of algorithmic bytes (x in, x' out).
//
// What makes one launch possible is that BOTH sides of the softmax are linear in things that do not depend on the step:
//     S_h  = LN(x) Wq_h^T K_h^T          = LN(x) (K_h Wq_h)^T          =: LN(x) A_h^T           A_h [keys, C]   (one matrix per head and context)
//     o Wo = sum_h P_h V_h Wo_h^T        = sum_h P_h (Wo_h V_h^T)^T    =: sum_h P_h B_h^T       B_h [C, keys]
// so a head is two small GEMMs against per-(context, head) matrices the host derives ONCE per sampling / inversion loop (xattn.py: fp32 products of the
// fp16 weights, one rounding -- the query and the attention output are never rounded to fp16 at all), head_dim disappears, and the C-wide output
// accumulators can stay in registers across the heads.  LayerNorm is folded as in gsw_gemm_ln: S = rstd (x A'^T) + nrm u + v with A' = A diag(gamma),
// softmax scale and log2(e) folded in, u = A' 1, v = A beta (fp32, -inf for the padding keys: the mask costs nothing).
//
// Kernel (C = 320, the 64 x 64 level of SD 2.1 and the 96 x 96 level of SD 1.5; <= 80 keys; any number of heads):
//   * one workgroup = 128 token rows = 4 waves x 32 rows, ONE wave per SIMD with the whole 512-register file: 160 accumulators (32 rows x 320 columns of x'),
//     the wave's rows of x as 20 B-operand fragments (80 registers, read from HBM once: they feed every head's first product AND the residual),
//     40 score accumulators
//   * every product is computed TRANSPOSED (weights = A operand from LDS, rows = B operand from registers, v_mfma_f32_16x16x32): the accumulator of
//     S^T holds, per lane, four keys of one row -- exactly the B-operand layout of the second product, so P never moves between lanes; the contraction
//     order over the keys (and the order of the output columns) is whatever that layout dictates and the host stores the matrices in it
//   * the residual and the output bias ride on the matrix pipe: x' accumulators start as (permutation matrix) x (the x fragments) -- exact in fp32 -- and the
//     bias is the row of B_h for a 81st key whose probability is the constant 1.0 (last head only)
//   * the per-(context, head) matrices arrive as a stream of 1 KiB MFMA fragments in consumption order (110 per head): global -> registers -> LDS in chunks of
//     ten, three chunks of lead in registers, a three-slot LDS ring, ONE barrier per chunk (20-24 MFMAs per wave); the stream runs across heads and tiles
//   * epilogue: one rounding, 16-byte stores, and the (rstd, -rstd mean) of the NEW rows for the LayerNorm that follows (norm3) -- the whole row is in the wave
//   * tiles of one image go to ONE XCD at a time (a context's matrices -- 550 KB -- are read from HBM once and shared through that XCD's L2)
// Roofline: HBM (0.67 GB per launch at 128 images) / LDS fragment reads (every fragment feeds two MFMAs: 50 % of the LDS read rate at full MFMA rate).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "../../include/gswm.h"
#include "gswm_mmtypes.h"

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip

namespace {

constexpr int XC = 320;                 // channels of the level this kernel serves
constexpr int XKS = XC / 32;            // k-steps of the first product
constexpr int XNB = XC / 16;            // 16-column blocks of the output
constexpr int XKB = 5;                  // 16-key blocks: 80 key slots
constexpr uint32_t XCHUNK = 10240;      // ten 1 KiB fragments
constexpr uint32_t XHEAD = 11 * XCHUNK; // one head: 50 fragments of A' (k-step major, key block minor) + 60 of B (key step major, column block minor)
constexpr int XUV = 2 * 16 * XKB;       // floats of (u | v) per head

struct XArgs {
    const uint16_t* x;      // [xB * S, 320] raw residual stream
    const float2* stat;     // [xB * S] (rstd, -rstd mean) of its rows
    const uint8_t* blob;    // per context: heads * XHEAD bytes of fragments (xattn.py: build_context_blob)
    const float* uv;        // per context: heads * XUV floats
    const int32_t* bidx;    // [oB] context of an output image (nullptr: all 0)
    uint16_t* out;          // [oB * S, 320]
    float2* ostat;          // [oB * S] (rstd, -rstd mean) of the output rows (nullptr: not wanted)
    int64_t blob_stride;    // bytes between contexts
    int64_t uv_stride;      // floats between contexts
    float inv_c, eps;
    uint32_t xB, oB, S, T;  // images of x (output image i reads x image i % xB), output images, tokens per image, 128-row tiles per image
    uint32_t heads, ntiles, xcd;
};

// rows 0-1 and 2-3 of the four 16-lane rows combined, then the two halves: every lane ends up with the reduction over lanes (l & 15) + 16 k
__device__ __forceinline__ float xq_max(float v) {
    uint32_t u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xq_sum(float v) {
    uint32_t u = __float_as_uint(v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// a wave-uniform 64-bit offset, said so: a global load then addresses (kernel-argument pointer + it) as an SGPR pair + a 32-bit lane offset instead of a 64-bit add per lane
__device__ __forceinline__ int64_t xuni(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// context of an output image, by an explicit SCALAR load: hipcc reads a uniform address in writable memory with a vector load and waits for it with vmcnt(0) -- at the
// top of a tile that is a wait for the previous tile's 40 output stores
__device__ __forceinline__ int64_t xctx(const int32_t* tab, uint32_t i) {
    const int32_t* q = tab + __builtin_amdgcn_readfirstlane(i);
    int32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
    return (int64_t)v;
}

// workgroup barrier behind this wave's LDS WRITES: LDS operations retire in order, so with the step's fragment reads issued behind its writes `lgkmcnt(N)` lets the N newest reads
// stay in flight across the barrier (a read still in flight here is done long before its slot is written again, two barriers later)
#define X_BARRIER(N) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// tile `it` of this workgroup -> (output image, 128-row tile inside it).  xcd: hardware block b runs on XCD b & 7; the 32 workgroups of an XCD walk the
// tiles of images xcd, xcd + 8, ... in order, so that one context's fragment stream is live in ONE L2 at a time.
__device__ __forceinline__ bool xtile(const XArgs& p, uint32_t it, uint32_t& oi, uint32_t& sub) {
    if (p.xcd) {
        const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, nper = gridDim.x >> 3;
        const uint32_t s = slot + it * nper, nimg = (p.oB - xcd + 7u) >> 3;
        if (s >= nimg * p.T) return false;
        const uint32_t q = s / p.T;
        oi = xcd + 8u * q;
        sub = s - q * p.T;
    } else {
        const uint32_t t = blockIdx.x + it * gridDim.x;
        if (t >= p.ntiles) return false;
        oi = t / p.T;
        sub = t - oi * p.T;
    }
    return true;
}

template <typename T>
__global__ __launch_bounds__(256) void gsw_xattn_kernel(const XArgs p) {
    using M_ = MM<T>;
    using frag = typename M_::frag;
    __shared__ __attribute__((aligned(16))) uint8_t ring[3 * XCHUNK];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t r = lane & 15u, g = lane >> 4;
    // this lane's share of a chunk: two whole fragments' 16 bytes and 8 bytes of a fragment shared with the neighbouring wave
    const uint32_t o0 = wave * 1024u + lane * 16u, o1 = o0 + 4096u, o2 = (8u + (wave >> 1)) * 1024u + (wave & 1u) * 512u + lane * 8u;
    constexpr uint32_t ONE = std::is_same<T, _Float16>::value ? 0x3C00u : 0x3F80u;      // 1.0 in the storage dtype
    // residual as a product: A[label][k] = 1 where k = 8 (label >> 2) + 4 e + (label & 3) picks column n(block 2 q + e, label) = 32 q + 8 (label >> 2) + 4 e + (label & 3) of k-step q
    frag pm[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const uint32_t idx = 4u * e + (r & 3u), one = g == (r >> 2) ? ONE << (16u * (idx & 1u)) : 0u;
        pm[e] = __builtin_bit_cast(frag, uint4{(idx >> 1) == 0u ? one : 0u, (idx >> 1) == 1u ? one : 0u, (idx >> 1) == 2u ? one : 0u, (idx >> 1) == 3u ? one : 0u});
    }
    const uint32_t pad_one = g == 0u ? ONE : 0u;          // key slot 80 (lane row 0, element 4 of the third key step): probability 1.0 for the bias row

    uint32_t oi, sub;
    if (!xtile(p, 0, oi, sub)) return;
    int64_t cctx = p.bidx ? xctx(p.bidx, oi) : 0;      // this tile's context
    const uint8_t* cur = p.blob + xuni(cctx * p.blob_stride);

    // three staging register sets, named (not an array: every use must be a compile-time choice for them to stay in registers)
    uint4 sa0, sa1, sa2, sb0, sb1, sb2;
    uint2 sc0, sc1, sc2;
#define X_LD(st, cp) do { const uint8_t* cp_ = (cp); const uint4 a_ = *reinterpret_cast<const uint4*>(cp_ + o0), b_ = *reinterpret_cast<const uint4*>(cp_ + o1); \
                          const uint2 c_ = *reinterpret_cast<const uint2*>(cp_ + o2);                                                                             \
                          if constexpr ((st) == 0) { sa0 = a_; sb0 = b_; sc0 = c_; } else if constexpr ((st) == 1) { sa1 = a_; sb1 = b_; sc1 = c_; } else { sa2 = a_; sb2 = b_; sc2 = c_; } } while (0)
#define X_WR(st, slot) do { uint8_t* sp_ = ring + (slot) * XCHUNK;                                                                                                \
                            *reinterpret_cast<uint4*>(sp_ + o0) = (st) == 0 ? sa0 : (st) == 1 ? sa1 : sa2; *reinterpret_cast<uint4*>(sp_ + o1) = (st) == 0 ? sb0 : (st) == 1 ? sb1 : sb2; \
                            *reinterpret_cast<uint2*>(sp_ + o2) = (st) == 0 ? sc0 : (st) == 1 ? sc1 : sc2; } while (0)
    // invariant at the top of step j: the ten fragments of chunk j are in registers (fr[j & 1]), chunk j + 1 is in LDS slot (j + 1) % 3 (written a step ago, published by
    // this step's barrier), chunks j + 2 .. j + 4 are in staging register sets (j + 2 .. j + 4) % 3
    frag fr[2][10];
    X_LD(0, cur);
    X_LD(1, cur + XCHUNK);
    X_LD(2, cur + 2 * XCHUNK);
    X_WR(0, 0);
    X_LD(0, cur + 3 * XCHUNK);
    X_WR(1, 1);
    X_LD(1, cur + 4 * XCHUNK);
    X_BARRIER(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) fr[0][i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(ring + lane * 16u + i * 1024));

    const mm_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the wave's 32 rows of x as B-operand fragments + their LayerNorm statistics.  They are loaded ONE TILE AHEAD: the last head's second product does not read them, so the
    // next tile's rows are requested right behind that head's softmax -- under its 120 MFMAs and in front of this tile's output stores -- instead of at the top of the tile
    frag xf[2][XKS];
    float2 st[2];
    auto load_x = [&](uint32_t oi_, uint32_t sub_) __attribute__((always_inline)) {
        const int64_t xrow = (int64_t)(oi_ % p.xB) * p.S + sub_ * 128u + wave * 32u + r;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const uint16_t* xr = p.x + (xrow + rb * 16) * XC + g * 8u;
#pragma unroll
            for (int ks = 0; ks < XKS; ++ks) xf[rb][ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(xr + ks * 32));
            st[rb] = p.stat[xrow + rb * 16];
        }
    };
    load_x(oi, sub);
    for (uint32_t it = 0;; ++it) {
        uint32_t noi = oi, nsub = sub;
        const bool more = xtile(p, it + 1, noi, nsub);
        const int64_t nctx = more ? (p.bidx ? xctx(p.bidx, noi) : 0) : cctx;      // the next tile's context: the fragment stream runs on into it
        const int64_t orow = (int64_t)oi * p.S + sub * 128u + wave * 32u + r;
        mm_f4 acc[XNB][2];
#pragma unroll
        for (int nb = 0; nb < XNB; ++nb)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[nb][rb] = M_::mma(pm[nb & 1], xf[rb][nb >> 1], zero4);

        auto head = [&](const uint32_t h, auto LAST) __attribute__((always_inline)) {
            const uint8_t* hb = p.blob + xuni(cctx * p.blob_stride + (int64_t)h * XHEAD);
            const uint8_t* hn = p.blob + xuni(h + 1 == p.heads ? nctx * p.blob_stride : cctx * p.blob_stride + (int64_t)(h + 1) * XHEAD);
            const float* huv = p.uv + xuni(cctx * p.uv_stride + h * XUV) + g * 4u;
            mm_f4 S[XKB][2];
            frag pf[3][2];
            float4 u4[XKB], v4[XKB];
            auto step = [&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                X_BARRIER(j != 5 ? 3 : 0);      // (the previous step read this chunk's fragments behind its writes -- unless this is the empty chunk)
                {   // chunk j + 2: staging registers -> LDS (its slot held chunk j - 1, whose fragments every wave had in registers before the previous barrier)
                    constexpr int m = (j + 2) % 12;
                    if constexpr (m != 5) X_WR(m % 3, m % 3);
                }
                {   // chunk j + 5: global -> the staging set just freed
                    constexpr int m = (j + 5) % 12;
                    if constexpr (m != 5) X_LD(m % 3, ((j + 5) >= 12 ? hn : hb) + (m < 5 ? m : m - 1) * XCHUNK);
                }
                {   // chunk j + 1 (written a step ago, published by this barrier): LDS -> the other fragment set, under this chunk's MFMAs
                    constexpr int m = (j + 1) % 12;
                    if constexpr (m != 5) {
                        const uint8_t* sl = ring + (m % 3) * XCHUNK + lane * 16u;
#pragma unroll
                        for (int i = 0; i < 10; ++i) fr[m & 1][i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
                    }
                }
                if constexpr (j == 3) {
#pragma unroll
                    for (int kb = 0; kb < XKB; ++kb) {
                        u4[kb] = *reinterpret_cast<const float4*>(huv + kb * 16);
                        v4[kb] = *reinterpret_cast<const float4*>(huv + 16 * XKB + kb * 16);
                    }
                }
                if constexpr (j < 5) {
                    // S^T[key block kb][rows] += A'[16 keys x 32 channels] x^T[32 channels x 16 rows], two k-steps per chunk
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        const int ks = 2 * j + k2;
#pragma unroll
                        for (int kb = 0; kb < XKB; ++kb) {
                            const frag a = fr[j & 1][k2 * XKB + kb];
#pragma unroll
                            for (int rb = 0; rb < 2; ++rb) S[kb][rb] = M_::mma(a, xf[rb][ks], ks == 0 ? zero4 : S[kb][rb]);
                        }
                    }
                } else if constexpr (j == 5) {
                    // softmax over the 80 key slots of a row: 20 per lane, 4 lanes per row
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb) {
                        float s[4 * XKB];
                        const float rstd = st[rb].x, nrm = st[rb].y;
#pragma unroll
                        for (int kb = 0; kb < XKB; ++kb) {
                            s[4 * kb + 0] = fmaf(rstd, S[kb][rb][0], fmaf(nrm, u4[kb].x, v4[kb].x));
                            s[4 * kb + 1] = fmaf(rstd, S[kb][rb][1], fmaf(nrm, u4[kb].y, v4[kb].y));
                            s[4 * kb + 2] = fmaf(rstd, S[kb][rb][2], fmaf(nrm, u4[kb].z, v4[kb].z));
                            s[4 * kb + 3] = fmaf(rstd, S[kb][rb][3], fmaf(nrm, u4[kb].w, v4[kb].w));
                        }
                        float mx = s[0];
#pragma unroll
                        for (int i = 1; i < 4 * XKB; ++i) mx = fmaxf(mx, s[i]);
                        mx = xq_max(mx);
                        float l = 0.f;
#pragma unroll
                        for (int i = 0; i < 4 * XKB; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mx); l += s[i]; }
                        const float inv = __builtin_amdgcn_rcpf(xq_sum(l));
#pragma unroll
                        for (int kk = 0; kk < 3; ++kk) {
                            uint32_t w[4];
                            w[0] = M_::cvt2(s[8 * kk + 0] * inv, s[8 * kk + 1] * inv);
                            w[1] = M_::cvt2(s[8 * kk + 2] * inv, s[8 * kk + 3] * inv);
                            if (kk < 2) {
                                w[2] = M_::cvt2(s[8 * kk + 4] * inv, s[8 * kk + 5] * inv);
                                w[3] = M_::cvt2(s[8 * kk + 6] * inv, s[8 * kk + 7] * inv);
                            } else {
                                w[2] = pad_one;
                                w[3] = 0u;
                            }
                            pf[kk][rb] = __builtin_bit_cast(frag, uint4{w[0], w[1], w[2], w[3]});
                        }
                    }
                    if constexpr (decltype(LAST)::value) load_x(noi, nsub);      // (the last tile re-reads its own rows: harmless)
                } else {
                    // x'^T[column block nb][rows] += B[16 columns x 32 key slots] P^T[32 key slots x 16 rows], ten column blocks per chunk
                    constexpr int kk = (j - 6) >> 1, nb0 = 10 * ((j - 6) & 1);
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const frag a = fr[j & 1][i];
#pragma unroll
                        for (int rb = 0; rb < 2; ++rb) acc[nb0 + i][rb] = M_::mma(a, pf[kk][rb], acc[nb0 + i][rb]);
                    }
                }
                // issue order inside the step (one wave per SIMD: what this wave does not overlap, nothing does): the LDS writes and the global loads first -- they complete
                // under the MFMAs instead of in front of the next barrier's wait --, then the MFMAs with the fragment reads of the next chunk between them
                if constexpr (j != 5) {
                    if constexpr ((j + 2) % 12 != 5) __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
                    if constexpr ((j + 5) % 12 != 5 || j == 3) __builtin_amdgcn_sched_group_barrier(0x020, ((j + 5) % 12 != 5 ? 3 : 0) + (j == 3 ? 2 * XKB : 0), 0);
                    if constexpr ((j + 1) % 12 != 5) {
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
                    }
                }
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        };
        for (uint32_t h = 0; h + 1 < p.heads; ++h) head(h, std::false_type{});
        head(p.heads - 1, std::true_type{});      // (its own copy of the code: the x loads of the next tile sit in it unconditionally)

        // epilogue: lane (r, g) holds columns 32 q + 8 g .. + 7 of rows r and 16 + r
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            float sm = 0.f, sq = 0.f;
            uint16_t* orp = p.out + (orow + rb * 16) * XC + g * 8u;
#pragma unroll
            for (int q = 0; q < XKS; ++q) {
                uint4 w;
                w.x = M_::cvt2(acc[2 * q][rb][0], acc[2 * q][rb][1]);
                w.y = M_::cvt2(acc[2 * q][rb][2], acc[2 * q][rb][3]);
                w.z = M_::cvt2(acc[2 * q + 1][rb][0], acc[2 * q + 1][rb][1]);
                w.w = M_::cvt2(acc[2 * q + 1][rb][2], acc[2 * q + 1][rb][3]);
                M_::stat2(w.x, sm, sq);
                M_::stat2(w.y, sm, sq);
                M_::stat2(w.z, sm, sq);
                M_::stat2(w.w, sm, sq);
                *reinterpret_cast<uint4*>(orp + q * 32) = w;
            }
            if (p.ostat) {
                sm = xq_sum(sm);
                sq = xq_sum(sq);
                const float mean = sm * p.inv_c;
                const float var = fmaxf(sq * p.inv_c - mean * mean, 0.f);
                const float rstd = rsqrtf(var + p.eps);
                if (g == 0u) p.ostat[orow + rb * 16] = make_float2(rstd, -rstd * mean);
            }
        }
        if (!more) break;
        oi = noi; sub = nsub; cctx = nctx;
    }
#undef X_LD
#undef X_WR
}

}  // namespace

int gsw_xattn_fused(const void* x_dev, const float* ln_stat_dev, const void* blob_dev, int64_t blob_stride_bytes, const float* uv_dev, int64_t uv_stride_floats,
                    const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images, int out_images, int tokens, int C, int heads,
                    int dtype, void* stream) {
    if (!x_dev || !ln_stat_dev || !blob_dev || !uv_dev || !out_dev || x_images <= 0 || out_images <= 0 || tokens <= 0 || heads <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (((uintptr_t)x_dev | (uintptr_t)blob_dev | (uintptr_t)uv_dev | (uintptr_t)out_dev | (uintptr_t)ln_stat_dev | (uintptr_t)out_stat_dev) & 15) return GSW_ERR_BAD_ARG;
    if ((blob_stride_bytes & 15) || (uv_stride_floats & 3) || out_images % x_images) return GSW_ERR_BAD_ARG;
    if (C != XC || tokens % 128 || heads > 64 || (int64_t)out_images * tokens >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        g_last_hip_error = (int)hipGetLastError();
        return GSW_ERR_HIP;
    }
    XArgs a;
    a.x = reinterpret_cast<const uint16_t*>(x_dev);
    a.stat = reinterpret_cast<const float2*>(ln_stat_dev);
    a.blob = reinterpret_cast<const uint8_t*>(blob_dev);
    a.uv = uv_dev;
    a.bidx = ctx_index_dev;
    a.out = reinterpret_cast<uint16_t*>(out_dev);
    a.ostat = reinterpret_cast<float2*>(out_stat_dev);
    a.blob_stride = blob_stride_bytes;
    a.uv_stride = uv_stride_floats;
    a.inv_c = 1.0f / (float)XC;
    a.eps = out_eps;
    a.xB = (uint32_t)x_images; a.oB = (uint32_t)out_images; a.S = (uint32_t)tokens; a.T = (uint32_t)(tokens / 128);
    a.heads = (uint32_t)heads;
    a.ntiles = a.oB * a.T;
    // XCD-ordered tiles when there is more than one round of work and at least one image per XCD; else tiles in plain order over as many workgroups as there are tiles
    a.xcd = (cus % 8 == 0 && a.oB >= 8u && a.ntiles > (uint32_t)cus) ? 1u : 0u;
    const uint32_t grid = a.xcd ? (uint32_t)cus : std::min<uint32_t>((uint32_t)cus, a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GSW_F16) hipLaunchKernelGGL(gsw_xattn_kernel<_Float16>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gsw_xattn_kernel<__bf16>, dim3(grid), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}
