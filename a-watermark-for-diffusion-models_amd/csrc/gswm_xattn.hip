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
// accumulators can stay in registers across the heads.  LayerNorm is folded: A' = A diag(gamma) with softmax scale and log2(e) folded in and every row CENTRED over
// the channels, so that x A'^T = (x - mean(x)) A'^T and S = rstd (x A'^T) + v, v = A beta (fp32, -inf for the padding keys: the mask costs nothing).
//
// Kernel (C = 320, the 64 x 64 level of SD 2.1 and the 96 x 96 level of SD 1.5; <= 79 keys; any number of heads):
//   * one workgroup = 128 token rows = 4 waves x 32 rows, ONE wave per SIMD with the whole 512-register file: 160 accumulators (32 rows x 320 columns of x'),
//     the wave's rows of x as 20 B-operand fragments (80 registers, read from HBM once: they feed every head's first product AND the residual),
//     48 score accumulators
//   * every product is computed TRANSPOSED (weights = A operand from LDS, rows = B operand from registers, v_mfma_f32_32x32x16: one MFMA per fragment read -- with one
//     wave per SIMD nothing but this wave's own instruction stream overlaps the matrix pipe, and the 16 x 16 x 32 form of round-6's first build had too many instructions
//     per MFMA to hide): the accumulator of S^T holds, per lane, keys of ONE row -- exactly the B-operand layout of the second product, so P never moves between lanes;
//     the contraction order over the keys (and the order of the output columns) is whatever that layout dictates and the host stores the matrices in it
//   * the residual and the output bias ride on the matrix pipe: x' accumulators start as (permutation matrix) x (the x fragments) -- exact in fp32 -- and the
//     bias is the row of B_h for key slot 79, whose probability is the constant 1.0 (last head only)
//   * the per-(context, head) matrices arrive as a stream of 1 KiB MFMA fragments in consumption order (60 + 50 per head): global -> registers -> LDS in chunks of
//     ten, three chunks of lead in registers, a three-slot LDS ring, ONE barrier per chunk (10 MFMAs of 32 cycles per wave); a chunk's fragments are read from LDS one
//     chunk ahead, each into the registers of the fragment the previous MFMA consumed; the stream runs across heads and tiles
//   * the rows of x are loaded one tile ahead (behind the last head's softmax, under its second product, in front of the output stores)
//   * epilogue: one rounding, 16-byte stores, and the (rstd, -rstd mean) of the NEW rows for the LayerNorm that follows (norm3) -- a row is two lanes of one wave
//   * tiles of one image go to ONE XCD at a time (a context's matrices -- 550 KB -- are read from HBM once and shared through that XCD's L2)
// Roofline: HBM (0.67 GB per launch at 128 images); measured and what bounds it: DESIGN.md section 4.9.
// The same transposed-stream machinery serves two more launches in this file: the PRO form of the kernel (the self-attention's output projection + residual + norm2 statistics
// as a prologue: gsw_xattn_fused_pre) and gsw_gnproj_kernel (GroupNorm + proj_in at the entry of a transformer: gsw_gn_proj_tokens); both are described where they are defined.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <type_traits>

#include "../../include/gswm.h"
#include "gswm_mmtypes.h"

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip

namespace {

typedef float x_f16v __attribute__((ext_vector_type(16)));

constexpr int XC = 320;                 // channels of the level this kernel serves
constexpr int XKS = XC / 16;            // 16-channel k-steps of the first product (v_mfma_f32_32x32x16)
constexpr int XNB = XC / 32;            // 32-column blocks of the output
constexpr int XKB = 3;                  // 32-key blocks of the first product: 96 key slots, of which the second product uses 80 (5 k-steps of 16)
constexpr int XKK = 5;
constexpr int XEMPTY = 6;               // the step of a head that carries the softmax instead of a chunk
constexpr uint32_t XCHUNK = 10240;      // ten 1 KiB fragments
constexpr int XPRO = 21;                 // chunks of the PRO kernels' prologue: 20 k-steps of Wo + the bias
constexpr uint32_t XHEAD = 11 * XCHUNK; // one head: 60 fragments of A' (k-step major, key block minor) + 50 of B (key step major, column block minor)
constexpr int XV = 32 * XKB;            // floats of v per head

struct XArgs {
    const uint16_t* x;      // [xB * S, 320] raw residual stream
    const float2* stat;     // [xB * S] (rstd, -rstd mean) of its rows
    const uint8_t* blob;    // per context: heads * XHEAD bytes of fragments (xattn.py: pack_stream)
    const float* uv;        // per context: heads * XV floats (v; -inf on padding keys)
    const int32_t* bidx;    // [oB] context of an output image (nullptr: all 0)
    uint16_t* out;          // [oB * S, 320]
    float2* ostat;          // [oB * S] (rstd, -rstd mean) of the output rows (nullptr: not wanted)
    const uint16_t* pre_o;  // PRO kernels: [xB * S, 320] the self-attention output; x is then the RESIDUAL of its output projection and stat is not read
    const uint8_t* pre_w;   // PRO kernels: 21 chunks -- 200 fragments of Wo (k-step major, column block minor) + ten bias fragments (xattn.py: pack_out_projection)
    int64_t blob_stride;    // bytes between contexts
    int64_t uv_stride;      // floats between contexts
    float inv_c, eps, eps_in;      // 1 / 320; epsilon of the LayerNorm behind the output (ostat) and -- PRO -- of the one in front (norm2)
    uint32_t xB, oB, S, T;  // images of x (output image i reads x image i % xB), output images, tokens per image, 128-row tiles per image
    uint32_t heads, ntiles, xcd;
};

template <typename T> struct XM;
template <> struct XM<_Float16> {
    static __device__ __forceinline__ x_f16v mma(mm_h8 a, mm_h8 b, x_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct XM<__bf16> {
    static __device__ __forceinline__ x_f16v mma(mm_b8 a, mm_b8 b, x_f16v c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// the value of the lane 32 positions away, combined (v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ float xh_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xh_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// a wave-uniform 64-bit offset, said so: a global load then addresses (kernel-argument pointer + it) as an SGPR pair + a 32-bit lane offset instead of a 64-bit add per lane
__device__ __forceinline__ int64_t xuni(int64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// context of an output image, by an explicit SCALAR load: hipcc reads a uniform address in writable memory with a vector load and waits for it with vmcnt(0) -- at the
// top of a tile that is a wait for the previous tile's 20 output stores
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

// PRO: the launch also runs what stands in front of the sublayer in a transformer block -- the output projection of the SELF-attention with its bias and residual,
// x = r + o Wo^T + b -- as a prologue per tile (x'^T accumulators = Wo fragments x the o fragments, 21 more chunks of stream per tile), takes norm2's statistics from the rows it
// has just made (a row is two lanes of one wave) and rounds them into the x fragments: x is never written or read, `gsw_ln_rowstats_finish` does not run
template <typename T, bool PRO>
__global__ __launch_bounds__(256) void gsw_xattn_kernel(const XArgs p) {
    using M_ = MM<T>;
    using frag = typename M_::frag;
    __shared__ __attribute__((aligned(16))) uint8_t ring0[XCHUNK], ring1[XCHUNK], ring2[XCHUNK];      // three slots, three objects: a slot's writes may pass another slot's reads
#define X_SLOT(k) ((k) == 0 ? ring0 : (k) == 1 ? ring1 : ring2)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t r = lane & 31u, hlf = lane >> 5;       // the lane's row of the wave's 32, and which 8 of a k-step's 16 values / which rows of an accumulator block it holds
    // this lane's share of a chunk: two whole fragments' 16 bytes and 8 bytes of a fragment shared with the neighbouring wave
    const uint32_t o0 = wave * 1024u + lane * 16u, o1 = o0 + 4096u, o2 = (8u + (wave >> 1)) * 1024u + (wave & 1u) * 512u + lane * 8u;
    constexpr uint32_t ONE = std::is_same<T, _Float16>::value ? 0x3C00u : 0x3F80u;      // 1.0 in the storage dtype
    // the residual as a product: accumulator row m of column block nb is output column 32 nb + 16 (m >> 4) + 8 ((m >> 2) & 1) + 4 ((m >> 3) & 1) + (m & 3), i.e. value
    // 8 ((m >> 2) & 1) + 4 ((m >> 3) & 1) + (m & 3) of k-step 2 nb + (m >> 4): A[m][k] = 1 there (a lane holds A[m = its row][8 hlf ..])
    frag pm[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t e = 4u * ((r >> 3) & 1u) + (r & 3u), one = ((r >> 4) == (uint32_t)j && ((r >> 2) & 1u) == hlf) ? ONE << (16u * (e & 1u)) : 0u;
        pm[j] = __builtin_bit_cast(frag, uint4{(e >> 1) == 0u ? one : 0u, (e >> 1) == 1u ? one : 0u, (e >> 1) == 2u ? one : 0u, (e >> 1) == 3u ? one : 0u});
    }

    uint32_t oi, sub;
    if (!xtile(p, 0, oi, sub)) return;
    int64_t cctx = p.bidx ? xctx(p.bidx, oi) : 0;      // this tile's context
    const uint8_t* cur = PRO ? p.pre_w : p.blob + xuni(cctx * p.blob_stride);      // where this workgroup's fragment stream starts

    // three staging register sets, named (not an array: every use must be a compile-time choice for them to stay in registers)
    uint4 sa0, sa1, sa2, sb0, sb1, sb2;
    uint2 sc0, sc1, sc2;
#define X_LD(st, cp) do { const uint8_t* cp_ = (cp); const uint4 a_ = *reinterpret_cast<const uint4*>(cp_ + o0), b_ = *reinterpret_cast<const uint4*>(cp_ + o1); \
                          const uint2 c_ = *reinterpret_cast<const uint2*>(cp_ + o2);                                                                             \
                          if constexpr ((st) == 0) { sa0 = a_; sb0 = b_; sc0 = c_; } else if constexpr ((st) == 1) { sa1 = a_; sb1 = b_; sc1 = c_; } else { sa2 = a_; sb2 = b_; sc2 = c_; } } while (0)
#define X_WR(st, slot) do { uint8_t* sp_ = X_SLOT(slot);                                                                                                \
                            *reinterpret_cast<uint4*>(sp_ + o0) = (st) == 0 ? sa0 : (st) == 1 ? sa1 : sa2; *reinterpret_cast<uint4*>(sp_ + o1) = (st) == 0 ? sb0 : (st) == 1 ? sb1 : sb2; \
                            *reinterpret_cast<uint2*>(sp_ + o2) = (st) == 0 ? sc0 : (st) == 1 ? sc1 : sc2; } while (0)
    // invariant at the top of step j: the ten fragments of chunk j are in registers (fr), chunk j + 1 is in LDS slot (j + 1) % 3 (written a step ago, published by
    // this step's barrier), chunks j + 2 .. j + 4 are in staging register sets (j + 2 .. j + 4) % 3
    frag fr[10];
    X_LD(0, cur);
    X_LD(1, cur + XCHUNK);
    X_LD(2, cur + 2 * XCHUNK);
    X_WR(0, 0);
    X_LD(0, cur + 3 * XCHUNK);
    X_WR(1, 1);
    X_LD(1, cur + 4 * XCHUNK);
    X_BARRIER(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(ring0 + lane * 16u + i * 1024));

    const x_f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the lane's row of x as B-operand fragments (lane (row, hlf) holds channels 16 ks + 8 hlf .. + 7 of k-step ks) + its LayerNorm statistics.  They are loaded ONE TILE
    // AHEAD: the last head's second product does not read them, so the next tile's rows are requested right behind that head's softmax -- under its 50 MFMAs and in front
    // of this tile's output stores -- instead of at the top of the tile
    frag xf[XKS];
    float2 st;
    frag of[XKS];            // PRO: the lane's row of the self-attention output, loaded one tile ahead like x in the plain kernel (dead while the heads run)
    auto load_x = [&](uint32_t oi_, uint32_t sub_) __attribute__((always_inline)) {
        const int64_t xrow = (int64_t)(oi_ % p.xB) * p.S + sub_ * 128u + wave * 32u + r;
        const uint16_t* xr = p.x + xrow * XC + hlf * 8u;
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) xf[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(xr + ks * 16));
        if constexpr (!PRO) st = p.stat[xrow];
    };
    auto load_o = [&](uint32_t oi_, uint32_t sub_) __attribute__((always_inline)) {
        const int64_t xrow = (int64_t)(oi_ % p.xB) * p.S + sub_ * 128u + wave * 32u + r;
        const uint16_t* orr = p.pre_o + xrow * XC + hlf * 8u;
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) of[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(orr + ks * 16));
    };
    if constexpr (PRO) load_o(oi, sub); else load_x(oi, sub);
    for (uint32_t it = 0;; ++it) {
        uint32_t noi = oi, nsub = sub;
        const bool more = xtile(p, it + 1, noi, nsub);
        const int64_t nctx = more ? (p.bidx ? xctx(p.bidx, noi) : 0) : cctx;      // the next tile's context: the fragment stream runs on into it
        const int64_t orow = (int64_t)oi * p.S + sub * 128u + wave * 32u + r;
        x_f16v acc[XNB];
        // (PRO: the prologue's stream sits at ONE address for every tile; an offset the compiler cannot see through keeps it from hoisting 63 per-lane 64-bit addresses out of
        //  the tile loop -- 126 registers this kernel does not have -- instead of adding a 32-bit lane offset to a scalar base at each load)
        int64_t opaque0 = 0;
        if constexpr (PRO) asm volatile("s_mov_b64 %0, 0" : "=s"(opaque0));
        const uint8_t* pw = PRO ? p.pre_w + opaque0 : nullptr;
        if constexpr (PRO) {
            // ---- prologue: x = r + o Wo^T + b in the x' accumulators; 21 steps of the same stream machinery (chunk j: the ten column blocks of k-step j; chunk 20: the bias) ----
            // (the residual r of the projection arrives as fragments in the second half of the prologue, two k-steps per step, into the registers the o fragments leave:
            //  o and r together never hold more than twenty fragments)
            const uint16_t* rr = p.x + ((int64_t)(oi % p.xB) * p.S + sub * 128u + wave * 32u + r) * XC + hlf * 8u;
            const uint8_t* h0 = p.blob + xuni(cctx * p.blob_stride);
            auto pstep = [&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                X_BARRIER(3);
                X_WR((j + 2) % 3, (j + 2) % 3);
                X_LD((j + 5) % 3, (j + 5) < XPRO ? pw + (j + 5) * XCHUNK : h0 + (j + 5 - XPRO) * XCHUNK);
                const uint8_t* sl = X_SLOT((j + 1) % 3) + lane * 16u;
                const frag e0 = __builtin_bit_cast(frag, uint4{hlf == 0u ? ONE : 0u, 0u, 0u, 0u});      // the bias fragments carry b in their k = 0 column
                constexpr bool rl = j >= XPRO - 11 && j < XPRO - 1;
                if constexpr (rl) {
                    xf[2 * (j - (XPRO - 11))] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(rr + (2 * (j - (XPRO - 11))) * 16));
                    xf[2 * (j - (XPRO - 11)) + 1] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(rr + (2 * (j - (XPRO - 11)) + 1) * 16));
                }
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    acc[i] = XM<T>::mma(fr[i], j < XPRO - 1 ? of[j < XPRO - 1 ? j : 0] : e0, j == 0 ? zero16 : acc[i]);
                    fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
                }
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (i >= 3 && i < (rl ? 8 : 6)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            };
            pstep(std::integral_constant<int, 0>{}); pstep(std::integral_constant<int, 1>{}); pstep(std::integral_constant<int, 2>{}); pstep(std::integral_constant<int, 3>{});
            pstep(std::integral_constant<int, 4>{}); pstep(std::integral_constant<int, 5>{}); pstep(std::integral_constant<int, 6>{}); pstep(std::integral_constant<int, 7>{});
            pstep(std::integral_constant<int, 8>{}); pstep(std::integral_constant<int, 9>{}); pstep(std::integral_constant<int, 10>{}); pstep(std::integral_constant<int, 11>{});
            pstep(std::integral_constant<int, 12>{}); pstep(std::integral_constant<int, 13>{}); pstep(std::integral_constant<int, 14>{}); pstep(std::integral_constant<int, 15>{});
            pstep(std::integral_constant<int, 16>{}); pstep(std::integral_constant<int, 17>{}); pstep(std::integral_constant<int, 18>{}); pstep(std::integral_constant<int, 19>{});
            pstep(std::integral_constant<int, 20>{});
            // + r (exact: permutation matrix x fragments), then ONE rounding: the rows of x as the three-launch path stores them, their statistics from the rounded values,
            // and the fragments both products below read
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int nb = 0; nb < XNB; ++nb) {
                acc[nb] = XM<T>::mma(pm[1], xf[2 * nb + 1], XM<T>::mma(pm[0], xf[2 * nb], acc[nb]));
            }
#pragma unroll
            for (int nb = 0; nb < XNB; ++nb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 w;
                    w.x = M_::cvt2(acc[nb][8 * j + 0], acc[nb][8 * j + 1]);
                    w.y = M_::cvt2(acc[nb][8 * j + 2], acc[nb][8 * j + 3]);
                    w.z = M_::cvt2(acc[nb][8 * j + 4], acc[nb][8 * j + 5]);
                    w.w = M_::cvt2(acc[nb][8 * j + 6], acc[nb][8 * j + 7]);
                    M_::stat2(w.x, sm, sq); M_::stat2(w.y, sm, sq); M_::stat2(w.z, sm, sq); M_::stat2(w.w, sm, sq);
                    xf[2 * nb + j] = __builtin_bit_cast(frag, w);
                }
            }
            sm = xh_sum(sm);
            sq = xh_sum(sq);
            const float mean = sm * p.inv_c, var = fmaxf(sq * p.inv_c - mean * mean, 0.f), rstd = rsqrtf(var + p.eps_in);
            st = make_float2(rstd, -rstd * mean);
        }
#pragma unroll
        for (int nb = 0; nb < XNB; ++nb) acc[nb] = XM<T>::mma(pm[1], xf[2 * nb + 1], XM<T>::mma(pm[0], xf[2 * nb], zero16));

        auto head = [&](const uint32_t h, auto LAST) __attribute__((always_inline)) {
            const uint8_t* hb = p.blob + xuni(cctx * p.blob_stride + (int64_t)h * XHEAD);
            const uint8_t* hn = (PRO && h + 1 == p.heads) ? pw      // (the next tile starts with the prologue's stream again)
                                                          : p.blob + xuni(h + 1 == p.heads ? nctx * p.blob_stride : cctx * p.blob_stride + (int64_t)(h + 1) * XHEAD);
            const float* hv = p.uv + xuni(cctx * p.uv_stride + h * XV) + hlf * 4u;
            x_f16v S[XKB];
            frag pf[XKK];
            float4 v4[XKK * 2];
            auto step = [&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value;
                X_BARRIER(j != XEMPTY ? 3 : 0);      // (the previous step read this chunk's fragments behind its writes -- unless this is the empty chunk)
                {   // chunk j + 2: staging registers -> LDS (its slot held chunk j - 1, whose fragments every wave had in registers before the previous barrier)
                    constexpr int m = (j + 2) % 12;
                    if constexpr (m != XEMPTY) X_WR(m % 3, m % 3);
                }
                {   // chunk j + 5: global -> the staging set just freed
                    constexpr int m = (j + 5) % 12;
                    if constexpr (m != XEMPTY) X_LD(m % 3, ((j + 5) >= 12 ? hn : hb) + (m < XEMPTY ? m : m - 1) * XCHUNK);
                }
                if constexpr (j == 4) {          // v of the 80 key slots the softmax looks at: two steps ahead of it
#pragma unroll
                    for (int q = 0; q < XKK * 2; ++q) v4[q] = *reinterpret_cast<const float4*>(hv + q * 8);
                }
                // this chunk's MFMAs; fragment i of chunk j + 1 (written a step ago, published by this barrier) is read from LDS into fragment i's registers behind the MFMA that
                // consumed them -- one chunk of lead, no second fragment set
                const uint8_t* sl = X_SLOT((j + 1) % 3) + lane * 16u;
                constexpr bool next_has = (j + 1) % 12 != XEMPTY;
                if constexpr (j < XEMPTY) {
                    // S^T[32 keys of block kb][rows] += A'[32 keys x 16 channels] x^T[16 channels x 32 rows]: fragment f = 3 ks + kb
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int f = 10 * j + i, ks = f / 3, kb = f % 3;
                        S[kb] = XM<T>::mma(fr[i], xf[ks], ks == 0 ? zero16 : S[kb]);
                        if constexpr (next_has) fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
                    }
                } else if constexpr (j == XEMPTY) {
                    // softmax over the 80 key slots of the lane's row: 40 in this lane (keys 32 kb + 8 (t >> 2) + 4 hlf + (t & 3) of accumulator element t), 40 in lane + 32
                    if constexpr (next_has) {
#pragma unroll
                        for (int i = 0; i < 10; ++i) fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
                    }
                    float s[8 * XKK];
#pragma unroll
                    for (int q = 0; q < XKK * 2; ++q) {
                        s[4 * q + 0] = fmaf(st.x, S[q >> 2][4 * (q & 3) + 0], v4[q].x);
                        s[4 * q + 1] = fmaf(st.x, S[q >> 2][4 * (q & 3) + 1], v4[q].y);
                        s[4 * q + 2] = fmaf(st.x, S[q >> 2][4 * (q & 3) + 2], v4[q].z);
                        s[4 * q + 3] = fmaf(st.x, S[q >> 2][4 * (q & 3) + 3], v4[q].w);
                    }
                    float mx = s[0];
#pragma unroll
                    for (int i = 1; i < 8 * XKK; ++i) mx = fmaxf(mx, s[i]);
                    mx = xh_max(mx);
                    float l = 0.f;
#pragma unroll
                    for (int i = 0; i < 8 * XKK; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mx); l += s[i]; }
                    const float inv = __builtin_amdgcn_rcpf(xh_sum(l));
#pragma unroll
                    for (int kk = 0; kk < XKK; ++kk) {
                        uint32_t w3 = M_::cvt2(s[8 * kk + 6] * inv, s[8 * kk + 7] * inv);
                        if (kk == XKK - 1 && hlf) w3 = (w3 & 0xFFFFu) | (ONE << 16);      // key slot 79 (upper lane half, last value of the fifth key step): probability 1.0 for the bias row
                        pf[kk] = __builtin_bit_cast(frag, uint4{M_::cvt2(s[8 * kk + 0] * inv, s[8 * kk + 1] * inv), M_::cvt2(s[8 * kk + 2] * inv, s[8 * kk + 3] * inv),
                                                                M_::cvt2(s[8 * kk + 4] * inv, s[8 * kk + 5] * inv), w3});
                    }
                    if constexpr (decltype(LAST)::value) { if constexpr (PRO) load_o(noi, nsub); else load_x(noi, nsub); }      // (the last tile re-reads its own rows: harmless)
                } else {
                    // x'^T[32 columns of block nb][rows] += B[32 columns x 16 key slots] P^T[16 key slots x 32 rows]: key step kk = j - 7, ten column blocks per chunk
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        acc[i] = XM<T>::mma(fr[i], pf[j - XEMPTY - 1], acc[i]);
                        if constexpr (next_has) fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
                    }
                }
                // issue order inside the step (one wave per SIMD: what this wave's own instruction stream does not put beside an MFMA, nothing overlaps -- measured in
                // isolation, tools/ubench/xattn_step.hip: the step's LDS writes, global loads, fragment reads and barrier are ~350 cycles next to 320 of MFMAs): every MFMA is
                // followed by one LDS write (first three gaps) / one global load (next three), then the fragment read into the registers it consumed.  The three LDS
                // operations issued last are reads, which is what X_BARRIER(3) relies on
                if constexpr (j != XEMPTY) {
                    constexpr bool wr = (j + 2) % 12 != XEMPTY, ld = (j + 5) % 12 != XEMPTY;
                    if constexpr (j == 4) __builtin_amdgcn_sched_group_barrier(0x020, 2 * XKK, 0);
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (wr && i < 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        if (ld && i >= 3 && i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        if constexpr (next_has) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        };
        for (uint32_t h = 0; h + 1 < p.heads; ++h) head(h, std::false_type{});
        head(p.heads - 1, std::true_type{});      // (its own copy of the code: the x loads of the next tile sit in it unconditionally)

        // epilogue: lane (row, hlf) holds columns 32 nb + 16 j + 8 hlf .. + 7 of its row in elements 8 j .. 8 j + 7 of block nb
        {
            float sm = 0.f, sq = 0.f;
            uint16_t* orp = p.out + orow * XC + hlf * 8u;
#pragma unroll
            for (int nb = 0; nb < XNB; ++nb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 w;
                    w.x = M_::cvt2(acc[nb][8 * j + 0], acc[nb][8 * j + 1]);
                    w.y = M_::cvt2(acc[nb][8 * j + 2], acc[nb][8 * j + 3]);
                    w.z = M_::cvt2(acc[nb][8 * j + 4], acc[nb][8 * j + 5]);
                    w.w = M_::cvt2(acc[nb][8 * j + 6], acc[nb][8 * j + 7]);
                    M_::stat2(w.x, sm, sq);
                    M_::stat2(w.y, sm, sq);
                    M_::stat2(w.z, sm, sq);
                    M_::stat2(w.w, sm, sq);
                    *reinterpret_cast<uint4*>(orp + nb * 32 + j * 16) = w;
                }
            }
            if (p.ostat) {
                sm = xh_sum(sm);
                sq = xh_sum(sq);
                const float mean = sm * p.inv_c;
                const float var = fmaxf(sq * p.inv_c - mean * mean, 0.f);
                const float rstd = rsqrtf(var + p.eps);
                if (hlf == 0u) p.ostat[orow] = make_float2(rstd, -rstd * mean);
            }
        }
        if (!more) break;
        oi = noi; sub = nsub; cctx = nctx;
    }
#undef X_LD
#undef X_WR
#undef X_SLOT
}


// ------------------------------------------------------------------------------------------------------------------------------------------------
// GroupNorm + proj_in of a Transformer2DModel at the 320-channel level as ONE launch: y = GroupNorm(x) Wp^T + b for every token, x a padded-flat (PF) NHWC
// tensor straight from the resnet in front (diffusers' Transformer2DModel.forward: `self.norm(hidden_states)` -> `proj_in`, behind extract.py:66-69).  As
// separate launches the normalised tokens were written by gsw_gn_pf_apply (356 MB in + 335 MB out at 128 images) and read back by a 320 x 320 GEMM that is pure
// traffic (335 in + 335 out); here the rows are normalised in registers on their way into the B operand and never stored.  The product is the prologue of the
// cross-attention kernel above on its own: the same 21-chunk fragment stream (xattn.py: pack_out_projection of proj_in), the same transposed accumulators,
// the same epilogue (one rounding, 16-byte stores, (rstd, -rstd mean) of the new rows for norm1).  GroupNorm's statistics come from the column records of the
// producing convolution as per-column-PAIR sums (gsw_gn_colstats_pairs); a workgroup folds them into per-channel (scale, shift) in LDS whenever its image changes
// (workgroups own CONTIGUOUS tile ranges: at most two images each at 128 images).
// Tried and dropped: the fragment stream by LDS-DMA (seven slots, six chunks of lead, inline-asm fragment reads and counted vmcnt waits: what hipcc needs to stay out of the
// way is in DESIGN.md section 8) -- correct, 213-215 us against 215-219 us: the stream is not what this kernel waits for (profiles/r06_gn_proj_fused_ab.txt).
// ------------------------------------------------------------------------------------------------------------------------------------------------
struct GPArgs {
    const uint16_t* x;        // PF rows [B][(H + 2)(W + 2)][320]
    const float2* pairsum;    // [B][160] (sum, sum of squares) of the channel pairs over the image's interior
    const uint16_t* gamma;    // [320]
    const uint16_t* beta;
    const uint8_t* w;         // 21 chunks (xattn.py: pack_out_projection)
    uint16_t* out;            // tokens [B * H * W][320]
    float2* ostat;            // [B * H * W] (rstd, -rstd mean) of the output rows (nullptr: not wanted)
    float inv_n, eps_gn, inv_c, eps_out;
    uint32_t B, H, W, Wp, HpWp, S, T, ntiles, cpg2;      // cpg2: channel pairs per group
};

template <typename T>
__global__ __launch_bounds__(256) void gsw_gnproj_kernel(const GPArgs p) {
    using M_ = MM<T>;
    using frag = typename M_::frag;
    __shared__ __attribute__((aligned(16))) uint8_t ring0[XCHUNK], ring1[XCHUNK], ring2[XCHUNK];
    __shared__ __attribute__((aligned(16))) float tab_sc[XC], tab_sh[XC];
#define G_SLOT(k) ((k) == 0 ? ring0 : (k) == 1 ? ring1 : ring2)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t r = lane & 31u, hlf = lane >> 5;
    const uint32_t o0 = wave * 1024u + lane * 16u, o1 = o0 + 4096u, o2 = (8u + (wave >> 1)) * 1024u + (wave & 1u) * 512u + lane * 8u;
    constexpr uint32_t ONE = std::is_same<T, _Float16>::value ? 0x3C00u : 0x3F80u;
    // this workgroup's contiguous range of tiles
    const uint32_t t_lo = (uint32_t)(((uint64_t)blockIdx.x * p.ntiles) / gridDim.x), t_hi = (uint32_t)(((uint64_t)(blockIdx.x + 1u) * p.ntiles) / gridDim.x);
    if (t_lo >= t_hi) return;

    uint4 sa0, sa1, sa2, sb0, sb1, sb2;
    uint2 sc0, sc1, sc2;
#define G_LD(st, cp) do { const uint8_t* cp_ = (cp); const uint4 a_ = *reinterpret_cast<const uint4*>(cp_ + o0), b_ = *reinterpret_cast<const uint4*>(cp_ + o1); \
                          const uint2 c_ = *reinterpret_cast<const uint2*>(cp_ + o2);                                                                             \
                          if constexpr ((st) == 0) { sa0 = a_; sb0 = b_; sc0 = c_; } else if constexpr ((st) == 1) { sa1 = a_; sb1 = b_; sc1 = c_; } else { sa2 = a_; sb2 = b_; sc2 = c_; } } while (0)
#define G_WR(st, slot) do { uint8_t* sp_ = G_SLOT(slot);                                                                                                \
                            *reinterpret_cast<uint4*>(sp_ + o0) = (st) == 0 ? sa0 : (st) == 1 ? sa1 : sa2; *reinterpret_cast<uint4*>(sp_ + o1) = (st) == 0 ? sb0 : (st) == 1 ? sb1 : sb2; \
                            *reinterpret_cast<uint2*>(sp_ + o2) = (st) == 0 ? sc0 : (st) == 1 ? sc1 : sc2; } while (0)
    frag fr[10];
    G_LD(0, p.w);
    G_LD(1, p.w + XCHUNK);
    G_LD(2, p.w + 2 * XCHUNK);
    G_WR(0, 0);
    G_LD(0, p.w + 3 * XCHUNK);
    G_WR(1, 1);
    G_LD(1, p.w + 4 * XCHUNK);
    X_BARRIER(0);
#pragma unroll
    for (int i = 0; i < 10; ++i) fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(ring0 + lane * 16u + i * 1024));

    const x_f16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // the lane's row of the tile: token -> PF row (y + 1) Wp + x + 1 of the image
    auto row_ptr = [&](uint32_t tile) __attribute__((always_inline)) {
        const uint32_t b = tile / p.T, tok = (tile - b * p.T) * 128u + wave * 32u + r;
        const uint32_t y = tok / p.W, xx = tok - y * p.W;
        return p.x + ((int64_t)b * p.HpWp + (int64_t)(y + 1u) * p.Wp + xx + 1u) * XC + hlf * 8u;
    };
    frag of[XKS];            // raw rows, normalised in place one step ahead of the MFMAs that read them
    {
        const uint16_t* xr = row_ptr(t_lo);
#pragma unroll
        for (int ks = 0; ks < XKS; ++ks) of[ks] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(xr + ks * 16));
    }
    // GroupNorm of fragment ks with the image's (scale, shift) table: fp32 arithmetic, ONE rounding (what gsw_gn_pf_apply stores)
    auto normalise = [&](frag& f, int ks) __attribute__((always_inline)) {
        const float4* ps = reinterpret_cast<const float4*>(tab_sc + ks * 16 + hlf * 8u);
        const float4* ph = reinterpret_cast<const float4*>(tab_sh + ks * 16 + hlf * 8u);
        const float4 s0 = ps[0], s1 = ps[1], h0 = ph[0], h1 = ph[1];
        const uint4 u = __builtin_bit_cast(uint4, f);
        uint4 w;
        w.x = M_::cvt2(fmaf(M_::up_lo(u.x), s0.x, h0.x), fmaf(M_::up_hi(u.x), s0.y, h0.y));
        w.y = M_::cvt2(fmaf(M_::up_lo(u.y), s0.z, h0.z), fmaf(M_::up_hi(u.y), s0.w, h0.w));
        w.z = M_::cvt2(fmaf(M_::up_lo(u.z), s1.x, h1.x), fmaf(M_::up_hi(u.z), s1.y, h1.y));
        w.w = M_::cvt2(fmaf(M_::up_lo(u.w), s1.z, h1.z), fmaf(M_::up_hi(u.w), s1.w, h1.w));
        f = __builtin_bit_cast(frag, w);
    };
    uint32_t cur_b = 0xFFFFFFFFu;
    for (uint32_t tile = t_lo; tile < t_hi; ++tile) {
        const uint32_t b = tile / p.T;
        if (b != cur_b) {      // (scale, shift) of the image's 320 channels from the pair sums of its groups
            cur_b = b;
            __syncthreads();
            if (threadIdx.x < XC / 2) {
                const uint32_t pr = threadIdx.x, g = pr / p.cpg2;
                const float2* q = p.pairsum + (int64_t)b * (XC / 2) + g * p.cpg2;
                float sm = 0.f, sq = 0.f;
                for (uint32_t k = 0; k < p.cpg2; ++k) { sm += q[k].x; sq += q[k].y; }
                const float mean = sm * p.inv_n, rstd = rsqrtf(fmaxf(sq * p.inv_n - mean * mean, 0.f) + p.eps_gn);
                const uint32_t gb = *reinterpret_cast<const uint32_t*>(p.gamma + 2u * pr), bb = *reinterpret_cast<const uint32_t*>(p.beta + 2u * pr);
                const float c0 = M_::up_lo(gb) * rstd, c1 = M_::up_hi(gb) * rstd;
                tab_sc[2u * pr] = c0; tab_sc[2u * pr + 1u] = c1;
                tab_sh[2u * pr] = M_::up_lo(bb) - mean * c0; tab_sh[2u * pr + 1u] = M_::up_hi(bb) - mean * c1;
            }
            __syncthreads();
        }
        const bool more = tile + 1u < t_hi;
        const uint16_t* xn = row_ptr(more ? tile + 1u : tile);      // (the last tile re-reads its own rows: harmless)
        const int64_t orow = (int64_t)tile * 128 + wave * 32u + r;      // tokens are dense: image b's tokens start at b S = b T 128
        normalise(of[0], 0);
        x_f16v acc[XNB];
        int64_t opaque0;      // (keeps hipcc from hoisting 63 loop-invariant 64-bit lane addresses of the stream out of the tile loop: see the cross-attention kernel)
        asm volatile("s_mov_b64 %0, 0" : "=s"(opaque0));
        const uint8_t* pw = p.w + opaque0;
        auto pstep = [&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            X_BARRIER(3);
            G_WR((j + 2) % 3, (j + 2) % 3);
            G_LD((j + 5) % 3, pw + ((j + 5) % XPRO) * XCHUNK);      // (the stream wraps into the next tile's)
            if constexpr (j + 1 < XPRO - 1) normalise(of[j + 1 < XKS ? j + 1 : 0], j + 1);
            const uint8_t* sl = G_SLOT((j + 1) % 3) + lane * 16u;
            const frag e0 = __builtin_bit_cast(frag, uint4{hlf == 0u ? ONE : 0u, 0u, 0u, 0u});
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                acc[i] = XM<T>::mma(fr[i], j < XPRO - 1 ? of[j < XPRO - 1 ? j : 0] : e0, j == 0 ? zero16 : acc[i]);
                fr[i] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(sl + i * 1024));
            }
            // the next tile's raw fragment j goes into the registers this step's MFMAs have just read
            if constexpr (j < XPRO - 1) of[j < XKS ? j : 0] = __builtin_bit_cast(frag, *reinterpret_cast<const uint4*>(xn + (j < XKS ? j : 0) * 16));
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                if (i >= 3 && i < 6) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        pstep(std::integral_constant<int, 0>{}); pstep(std::integral_constant<int, 1>{}); pstep(std::integral_constant<int, 2>{}); pstep(std::integral_constant<int, 3>{});
        pstep(std::integral_constant<int, 4>{}); pstep(std::integral_constant<int, 5>{}); pstep(std::integral_constant<int, 6>{}); pstep(std::integral_constant<int, 7>{});
        pstep(std::integral_constant<int, 8>{}); pstep(std::integral_constant<int, 9>{}); pstep(std::integral_constant<int, 10>{}); pstep(std::integral_constant<int, 11>{});
        pstep(std::integral_constant<int, 12>{}); pstep(std::integral_constant<int, 13>{}); pstep(std::integral_constant<int, 14>{}); pstep(std::integral_constant<int, 15>{});
        pstep(std::integral_constant<int, 16>{}); pstep(std::integral_constant<int, 17>{}); pstep(std::integral_constant<int, 18>{}); pstep(std::integral_constant<int, 19>{});
        pstep(std::integral_constant<int, 20>{});
        {
            float sm = 0.f, sq = 0.f;
            uint16_t* orp = p.out + orow * XC + hlf * 8u;
#pragma unroll
            for (int nb = 0; nb < XNB; ++nb) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    uint4 w;
                    w.x = M_::cvt2(acc[nb][8 * j + 0], acc[nb][8 * j + 1]);
                    w.y = M_::cvt2(acc[nb][8 * j + 2], acc[nb][8 * j + 3]);
                    w.z = M_::cvt2(acc[nb][8 * j + 4], acc[nb][8 * j + 5]);
                    w.w = M_::cvt2(acc[nb][8 * j + 6], acc[nb][8 * j + 7]);
                    M_::stat2(w.x, sm, sq);
                    M_::stat2(w.y, sm, sq);
                    M_::stat2(w.z, sm, sq);
                    M_::stat2(w.w, sm, sq);
                    *reinterpret_cast<uint4*>(orp + nb * 32 + j * 16) = w;
                }
            }
            if (p.ostat) {
                sm = xh_sum(sm);
                sq = xh_sum(sq);
                const float mean = sm * p.inv_c;
                const float var = fmaxf(sq * p.inv_c - mean * mean, 0.f);
                const float rstd = rsqrtf(var + p.eps_out);
                if (hlf == 0u) p.ostat[orow] = make_float2(rstd, -rstd * mean);
            }
        }
    }
#undef G_LD
#undef G_WR
#undef G_SLOT
}

}  // namespace

namespace {

int xattn_launch(const void* x_dev, const float* ln_stat_dev, const void* pre_o_dev, const void* pre_w_dev, float pre_eps, const void* blob_dev, int64_t blob_stride_bytes,
                 const float* v_dev, int64_t v_stride_floats, const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images, int out_images,
                 int tokens, int C, int heads, int dtype, void* stream) {
    const bool pro = pre_o_dev != nullptr;
    if (!x_dev || (!pro && !ln_stat_dev) || (pro && !pre_w_dev) || !blob_dev || !v_dev || !out_dev || x_images <= 0 || out_images <= 0 || tokens <= 0 || heads <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (((uintptr_t)x_dev | (uintptr_t)blob_dev | (uintptr_t)v_dev | (uintptr_t)out_dev | (uintptr_t)ln_stat_dev | (uintptr_t)out_stat_dev | (uintptr_t)pre_o_dev | (uintptr_t)pre_w_dev) & 15)
        return GSW_ERR_BAD_ARG;
    if ((blob_stride_bytes & 15) || (v_stride_floats & 3) || out_images % x_images) return GSW_ERR_BAD_ARG;
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
    a.uv = v_dev;
    a.bidx = ctx_index_dev;
    a.out = reinterpret_cast<uint16_t*>(out_dev);
    a.ostat = reinterpret_cast<float2*>(out_stat_dev);
    a.pre_o = reinterpret_cast<const uint16_t*>(pre_o_dev);
    a.pre_w = reinterpret_cast<const uint8_t*>(pre_w_dev);
    a.blob_stride = blob_stride_bytes;
    a.uv_stride = v_stride_floats;
    a.inv_c = 1.0f / (float)XC;
    a.eps = out_eps;
    a.eps_in = pre_eps;
    a.xB = (uint32_t)x_images; a.oB = (uint32_t)out_images; a.S = (uint32_t)tokens; a.T = (uint32_t)(tokens / 128);
    a.heads = (uint32_t)heads;
    a.ntiles = a.oB * a.T;
    // XCD-ordered tiles when there is more than one round of work and at least one image per XCD; else tiles in plain order over as many workgroups as there are tiles
    a.xcd = (cus % 8 == 0 && a.oB >= 8u && a.ntiles > (uint32_t)cus) ? 1u : 0u;
    const uint32_t grid = a.xcd ? (uint32_t)cus : std::min<uint32_t>((uint32_t)cus, a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GSW_F16) {
        if (pro) hipLaunchKernelGGL((gsw_xattn_kernel<_Float16, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gsw_xattn_kernel<_Float16, false>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (pro) hipLaunchKernelGGL((gsw_xattn_kernel<__bf16, true>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gsw_xattn_kernel<__bf16, false>), dim3(grid), dim3(256), 0, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

}  // namespace

int gsw_xattn_fused(const void* x_dev, const float* ln_stat_dev, const void* blob_dev, int64_t blob_stride_bytes, const float* v_dev, int64_t v_stride_floats,
                    const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images, int out_images, int tokens, int C, int heads,
                    int dtype, void* stream) {
    if (!ln_stat_dev) return GSW_ERR_BAD_ARG;
    return xattn_launch(x_dev, ln_stat_dev, nullptr, nullptr, 0.f, blob_dev, blob_stride_bytes, v_dev, v_stride_floats, ctx_index_dev, out_dev, out_stat_dev, out_eps, x_images,
                        out_images, tokens, C, heads, dtype, stream);
}

int gsw_xattn_fused_pre(const void* resid_dev, const void* o_dev, const void* w_frag_dev, float ln_eps, const void* blob_dev, int64_t blob_stride_bytes, const float* v_dev,
                        int64_t v_stride_floats, const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images, int out_images, int tokens,
                        int C, int heads, int dtype, void* stream) {
    if (!o_dev || !w_frag_dev) return GSW_ERR_BAD_ARG;
    return xattn_launch(resid_dev, nullptr, o_dev, w_frag_dev, ln_eps, blob_dev, blob_stride_bytes, v_dev, v_stride_floats, ctx_index_dev, out_dev, out_stat_dev, out_eps, x_images,
                        out_images, tokens, C, heads, dtype, stream);
}

int gsw_gn_proj_tokens(const void* x_pf_dev, const float* pairsum_dev, const void* gamma_dev, const void* beta_dev, float gn_eps, int groups, const void* w_frag_dev,
                       void* out_dev, float* out_stat_dev, float out_eps, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x_pf_dev || !pairsum_dev || !gamma_dev || !beta_dev || !w_frag_dev || !out_dev || B <= 0 || H <= 0 || W <= 0 || groups <= 0) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (((uintptr_t)x_pf_dev | (uintptr_t)pairsum_dev | (uintptr_t)w_frag_dev | (uintptr_t)out_dev | (uintptr_t)out_stat_dev) & 15) return GSW_ERR_BAD_ARG;
    if (((uintptr_t)gamma_dev | (uintptr_t)beta_dev) & 3) return GSW_ERR_BAD_ARG;
    // whole 128-token tiles per image, a wave's 32 tokens inside one image row, channel pairs inside one group
    if (C != XC || W % 32 || ((int64_t)H * W) % 128 || XC % groups || ((XC / groups) & 1) || (int64_t)B * H * W >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        g_last_hip_error = (int)hipGetLastError();
        return GSW_ERR_HIP;
    }
    GPArgs a;
    a.x = reinterpret_cast<const uint16_t*>(x_pf_dev);
    a.pairsum = reinterpret_cast<const float2*>(pairsum_dev);
    a.gamma = reinterpret_cast<const uint16_t*>(gamma_dev);
    a.beta = reinterpret_cast<const uint16_t*>(beta_dev);
    a.w = reinterpret_cast<const uint8_t*>(w_frag_dev);
    a.out = reinterpret_cast<uint16_t*>(out_dev);
    a.ostat = reinterpret_cast<float2*>(out_stat_dev);
    a.cpg2 = (uint32_t)(XC / groups / 2);
    a.inv_n = 1.0f / (float)((int64_t)H * W * (XC / groups));
    a.eps_gn = gn_eps;
    a.inv_c = 1.0f / (float)XC;
    a.eps_out = out_eps;
    a.B = (uint32_t)B; a.H = (uint32_t)H; a.W = (uint32_t)W; a.Wp = (uint32_t)(W + 2); a.HpWp = (uint32_t)((H + 2) * (W + 2));
    a.S = (uint32_t)(H * W); a.T = a.S / 128u; a.ntiles = a.B * a.T;
    const uint32_t grid = std::min<uint32_t>((uint32_t)cus, a.ntiles);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == GSW_F16) hipLaunchKernelGGL(gsw_gnproj_kernel<_Float16>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(gsw_gnproj_kernel<__bf16>, dim3(grid), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}
