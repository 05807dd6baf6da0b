// gswm_attn.hip -- self-attention of the eps-model (rows X2 / G1 of SURVEY.md section 8a) as a hand-written flash-attention
// forward for gfx950: softmax(Q K^T / sqrt(d)) V with head_dim 64, fp16 / bf16 operands, fp32 accumulation, non-causal.
//
// Reference call site: extract.py:66-69 / the generation loop run diffusers' UNet2DConditionModel, whose BasicTransformerBlock
// self-attention (attn1) at 64x64 / 32x32 / 16x16 latent resolution is sequence length 4096 / 1024 / 256 with 5 / 10 / 20 heads of
// width 64 (SD 2.1-base).  ~16 % of the UNet forward.
//
// Design (wave64, v_mfma_f32_32x32x16):
//   * one workgroup = 128 queries of one (batch, head): 4 waves x 32 queries; K and V tiles of 64 keys are staged once per
//     workgroup in LDS (double-buffered, register-prefetched) and shared by the 4 waves
//   * scores are computed TRANSPOSED, S^T = K Q^T (A = K tile rows, B = Q^T held in 16 VGPRs for the whole kernel): the accumulator
//     layout then puts one QUERY per lane column, so the softmax row reductions are lane-local plus one cross-half exchange, and the
//     probabilities are already in the B-operand layout of the second product O^T = V^T P^T -- no LDS round trip for P
//   * V is consumed as V^T [head_dim][keys] (the host computes the value projection transposed: it is one GEMM either way), so the
//     A operand of the second product is two 8-byte LDS reads per MFMA instead of a transposing gather
//   * the key order inside a 16-wide MFMA k-slice is the permutation the S^T accumulator layout dictates (slots 0-3 | 4-7 | 8-11 |
//     12-15 <-> keys 0-3 | 8-11 | 4-7 | 12-15); contraction order is free as long as V^T uses the same one
//   * LDS row pitches 144 B (K, ds_read_b128; 112 / 176 B for head_dim 40 / 80) and 136 B (V^T, ds_read_b64) are conflict-free for those
//     access widths
//   * head_dim 40 and 80 (the SD 1.5 shape: 8 heads at every level) run the same code with zero-padded k-slices / output row blocks
//   * head_dim 64 from 1024 keys up (round 6, template DMA): the K / V^T tiles go global -> LDS by DMA into four static stage arrays, unpadded swizzled rows, K rows
//     permuted inside every 16 so that V^T stays key-ordered -- no staging registers, no ds_write (see the comment in front of the kernel).  Same box, 128 images:
//     4096 keys 3.264 -> 3.216 ms, 1024 keys 0.488 -> 0.480 ms (power-capped shapes: -1.5 %; profiles/r06_attention_dma_ab.txt); GSW_ATTN_DMA=0 selects the register staging
//   * blockIdx -> (batch*head, query tile) is XCD-aware: the query tiles of one (batch, head) land on one XCD and share its L2 copy of K / V
// Roofline: MFMA.  Measured (B=128, 5 heads, S=4096, fp16): 825-832 TFLOP/s = 33 % of the 2.5 PF nominal peak, MFMA pipe busy 40-42 % of
// the cycles the chip actually runs (PMC: effective clock 1.87 GHz under this load); torch SDPA (aotriton) does 630-690 on the same shape.
// At head_dim 64 every 16 MFMAs come with ~170 VALU instructions (32 v_exp_f32, max / sum / convert / rescale) per wave.
// Row sums of P on the matrix pipe.  Round 1 tried one more 32 x 32 MFMA per 16-key slice with an all-ones A operand instead of the 32 v_add_f32 per tile and query
// block and dropped it (813 vs 831 TFLOP/s at 4096 keys: 2048 more matrix-pipe cycles per iteration).  Round 6 does it with instructions that cost next to nothing
// (LM / L4 in the kernel): at head_dim 40 V^T row 40 -- one of the eight rows of the 16-row tail block the head does not use -- is all ones in LDS, so the row sum is row
// 40 of the tail accumulators for free; every other form adds the lane's own four probabilities with v_mfma_f32_4x4x4 (sixteen independent 4 x 4 x 4 blocks, A = ones:
// 8 passes per tile and query block).  The sums are then those of the ROUNDED probabilities the second product multiplies.  What this removes is less the 64 VALU
// instructions than a serial chain: every v_add waited for its exponential.  Same box, old | new (profiles/r06_attention_rowsum_on_mfma_ab.txt): head_dim 40, 9216 keys
// 5.32-5.42 -> 4.88-5.05 ms (646 -> 700 algorithmic TFLOP/s: this shape is not power-capped), head_dim 80 0.589 -> 0.557 ms, head_dim 64 at 4096 keys (power-capped)
// 3.21-3.23 -> 3.17 ms.  Packing (s cs - m) as v_pk_fma_f32 on top: no change (measured again this round, as in round 4); the running maximum as four
// independent v_max3 chains instead of one: 1 % slower; ONE rescale branch per tile behind the softmax arithmetic of both query blocks instead of one per block
// (so that the blocks' instruction streams can interleave): no change.
// What sets the time (round 4, profiles/r04h_power_cap_probe.txt): on random operands the kernel runs at 1.99 GHz / 1.36 kW of the board's 1.4 kW, on all-zero operands
// at 2.39 GHz and finishes 28 % sooner -- the power management, not an issue port.  Measured in round 4 with ablation / A-B builds of this kernel (parts of the loop
// compiled out; packed v_pk_fma_f32 / v_pk_add_f32 softmax arithmetic and row sums by v_dot2c_f32_f16: 16-18 % fewer VALU instructions per tile, same time to the
// percent; s_setprio around the MFMA blocks: 3-4 % slower).  Those switches were removed from the product source in round 5: the builds are reproducible from the
// round-4 tree (git: 81b654b, tools/attn_ablate.sh there), their results are profiles/r04g_* and profiles/r04h_*.
// Tried and dropped: software pipelining over 32-key blocks (S^T of block u+1 issued before the softmax of block u, 3 LDS stages,
// one barrier per tile): 790 vs 831 TFLOP/s -- the per-block max / exchange / rescale overhead doubles and hipcc does not interleave
// the two streams any better than the wave scheduler already does across the 2-4 resident waves.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "../../include/gswm.h"

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// value of the lane 32 positions away, combined: v_permlane32_swap_b32 exchanges the upper half of one register with the lower half of
// another in the VALU (no LDS round trip like ds_bpermute, which sits on the critical path between the S^T MFMAs and the exponentials)
__device__ __forceinline__ float xhalf_max(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// (x of the lane's query for queries 0-15 | for queries 16-31 of the 32-query block) in EVERY 16-lane row: the 16 x 16 MFMA tail keeps one query per lane of a
// row, the 32 x 32 blocks one query per lane of a half -- one v_permlane16_swap moves a per-query scalar from the second layout to the first
__device__ __forceinline__ void rows16_split(float x, float& lo16, float& hi16) {
    const uint32_t u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);       // [0]: rows (0, 0, 2, 2) of x, [1]: rows (1, 1, 3, 3)
    lo16 = __uint_as_float(r[0]);
    hi16 = __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhalf_sum(float v) {
    const uint32_t u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename T> struct AT;
template <> struct AT<_Float16> {
    using v8 = f16x8;
    using v4 = f16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    // sixteen 4 x 4 x 4 blocks, one per four lanes: with A = ones every lane gets the sum of ITS OWN four B values in all four outputs (no cross-lane mixing)
    static __device__ __forceinline__ f32x4 mfma4(v4 a, v4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
};
template <> struct AT<__bf16> {
    using v8 = bf16x8;
    using v4 = bf16x4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x4 mfma4(v4 a, v4 b, f32x4 c) {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
};

struct AttnArgs {
    const void* q;      // [B, Sq, ...] row stride ldq elements; head h at column h*64
    const void* k;      // [B, Sk, ...] row stride ldk
    const void* vt;     // [B, H*64, Sk]: V transposed
    void* o;            // [B, Sq, ...] row stride ldo
    int32_t H, Sq, Sk, ldq, ldk, ldo;
    int32_t Sk_valid;   // keys >= Sk_valid are padding (masked out); == Sk when there is none
    float scale_log2;   // softmax scale * log2(e)
    uint32_t nqt;       // query tiles per (batch, head)
    uint32_t total;     // nqt * B * H
    uint32_t ksplit;    // KVS kernels: workgroups that share a query tile, each over its own range of key tiles (1 elsewhere)
    float* part;        // KVS kernels: partial results [total][ksplit][4 waves][NACC + 2][64 lanes] floats (unnormalised O^T accumulators, running maximum, partial row sum)
};

constexpr uint32_t VP = 144;                               // V^T LDS row pitch in bytes (64 keys + 16 B: an odd number of 16-byte slots -> conflict-free ds_read_b128)
// V^T rows are stored with the four 4-key groups of every 16 keys in the order (0, 2, 1, 3): the 16-wide MFMA k-slice of the second product wants keys
// {0-3, 8-11} in lanes 0-31 and {4-7, 12-15} in lanes 32-63 (the S^T accumulator layout), and in that order each half is ONE 16-byte read (round 4 read it as two
// ds_read_b64 from key-ordered rows: 16 instead of 8 V^T fragment reads per 64-key tile).  The permutation costs the staging nothing: it already wrote a 16-byte unit
// as two 8-byte halves (rows are only 8-byte aligned for them at the old pitch), now 16 bytes apart.

// QB = 32-query blocks per wave (1: 128 queries per workgroup; 2: 256 -- every K / V^T fragment read from LDS feeds two MFMAs and
// the two blocks' softmax / MFMA chains are independent instruction streams the scheduler can interleave inside one wave).
// DU = head_dim / 8 (5, 8, 10 <-> head_dim 40, 64, 80): the contraction over head_dim runs in KC = ceil(DU/2) MFMA k-slices and the
// output in DBF = head_dim / 32 row blocks of 32 (v_mfma_f32_32x32x16) plus, for head_dim 40 / 80, ONE block of 16 rows on v_mfma_f32_16x16x32 (TAIL): rows 32-39
// (+ 8 zero rows) / 64-79.  Round 4 ran the remainder as a full 32-row block -- at head_dim 40 that was 60 % more second-product MFMAs than the head has rows
// (SD 1.5 at 768 x 768: 9216-token self-attention, a third of that forward).  The tail's B operand (P^T with one QUERY per lane of a 16-lane row) is the S^T
// accumulator layout (one query per lane of a 32-lane half) after four v_permlane16_swap per 32 keys; the key order inside its 32-wide k-slice is what those swaps
// produce (keys 0-3 8-11 | 16-19 24-27 | 4-7 12-15 | 20-23 28-31 for the four lane rows) and the V^T fragment is read in the same order.  The padding lanes of Q are
// zero registers and the padding rows of V^T zero LDS rows.
// (A cross-tile software pipeline of the QB = 1 form -- the S^T MFMAs of tile t + 1 issued in front of the softmax arithmetic of tile t, three LDS stages -- was
// written in round 4, bit-identical and 3 % SLOWER at 4096 keys (3.74 vs 3.63 ms; profiles/r04g_attention_ablation_and_isa_mix.txt); removed in round 5.)
// PAIR: four LDS stages and ONE barrier per two 64-key tiles (half the barriers; 4 x 17.5 KiB of dynamic LDS at head_dim 64).
// RAGGED: Sq is not a multiple of the workgroup tile and / or Sk not a multiple of 64 (the mid block: 64 tokens at 512x512, 144 at 768x768;
// 576 tokens at the SD 1.5 third level).  Query lanes past Sq read row 0 and store nothing; key rows / V^T columns past Sk are fetched
// from a clamped (valid, finite) address and masked through Sk_valid like padded context keys.
// KVS (QB = 1, whole tiles; one image's self-attention: 160 workgroups for 256 CUs, each walking 64 key tiles): `ksplit` workgroups share a query tile, each over its
// own range of key tiles; they leave their unnormalised accumulators, running maximum and partial row sum in a workspace and gsw_attn_combine_kernel merges them
// (the usual rescaling by 2^(m_s - m)).  One image at 64 x 64: 62 -> ~30 us per launch.
// DMA (head_dim 64, PAIR, whole tiles, a multiple of four key tiles): K and V^T tiles go global -> LDS by DMA (global_load_lds, 16 bytes per lane, 1 KiB = eight 128-byte rows
// per wave instruction) into FOUR STATIC stage arrays -- no staging registers, no ds_write, no address arithmetic in the loop; separate __shared__ objects let hipcc's waitcnt pass
// see that a DMA into one stage cannot touch the stage being read (one dynamic array makes it wait vmcnt(0) in front of every fragment read).  Rows are unpadded; the 16-byte
// chunks of a row are XOR-swizzled with (row >> 1) & 7 on the DMA SOURCE address, which keeps every ds_read_b128 fragment read conflict-free (the matmul engine's scheme).  The
// 8-byte interleave of V^T rows that the register staging produced is replaced by a permutation of the K ROWS inside every 16 (LDS row 16 g + 8 a + 4 h + j holds key
// 16 g + 8 h + 4 a + j, again only a source address): the probabilities of a lane half h then belong to eight CONSECUTIVE keys and a V^T fragment is one chunk of a key-ordered row.
template <typename T, int QB, int DU, bool PAIR, bool RAGGED, bool KVS = false, bool DMA = false>
__global__ __launch_bounds__(256, (DU > 10 ? 1 : 2)) void gsw_attn_fwd_kernel(AttnArgs p) {
    using v8 = typename AT<T>::v8;
    using v4 = typename AT<T>::v4;
    constexpr int D = DU * 8, KC = (DU + 1) / 2, DBF = D / 32, TR = D % 32;
    constexpr bool TAIL = TR != 0;                                // head_dim 40 / 80: the last 8 / 16 output rows on 16 x 16 x 32 MFMAs
    static_assert(TR == 0 || TR == 8 || TR == 16, "head_dim % 32 must be 0, 8 or 16");
    constexpr int DB = DBF > 0 ? DBF : 1, VROWS = DBF * 32 + (TAIL ? 16 : 0), NACC = DBF * 16 + (TAIL ? 8 : 0);
    constexpr uint32_t KP = KC * 32 + 16;                         // K LDS row pitch: odd number of 16-byte slots -> conflict-free ds_read_b128
    constexpr uint32_t STAGE = 64 * KP + VROWS * VP;              // one 64-key K tile + one V^T tile
    constexpr int NU = (64 * DU + 255) / 256;                     // 16-byte staging units per thread, tile and operand
    constexpr uint32_t NSTG = PAIR ? 4 : 2;
    // LM (head_dim 40): the softmax row sum comes out of the matrix pipe.  The 16-row tail block has eight rows the head does not use; V^T row 40 is ALL ONES in LDS
    // (written once, the staging never touches padding rows), so row 40 of the tail accumulators is sum_k P[q][k] -- rescaled by alpha with the other rows, taken from the
    // ROUNDED probabilities the second product multiplies -- and the 64 v_add_f32 per tile and query block that kept it on the VALU are gone
    constexpr bool LM = TR == 8 && !KVS;
    // L4 (every other form): the lane's partial row sum on v_mfma_f32_4x4x4 -- sixteen independent 4 x 4 x 4 blocks, one per four lanes, A = ones: each lane gets the sum of
    // its own four ROUNDED probabilities added to its accumulator; eight of them (64 matrix-pipe cycles) replace 32 v_add_f32 (128 VALU cycles) per tile and query block
    constexpr bool L4 = !LM;
    static_assert(!KVS || (QB == 1 && !RAGGED && !PAIR), "key-split form: 32 queries per wave, whole tiles, the plain two-stage loop");
    static_assert(!DMA || (DU == 8 && PAIR && !RAGGED && !KVS), "DMA staging: head_dim 64, four stages, whole tiles");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];          // NSTG * STAGE bytes
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, h = lane >> 5, c32 = lane & 31u;
    constexpr uint32_t QW = 32u * QB, QWG = 4u * QW;          // queries per wave / per workgroup

    // XCD-aware placement: hardware sends workgroup b to XCD b % 8; give each XCD a contiguous range of logical ids, whose
    // consecutive members are the query tiles of one (batch, head)
    uint32_t logical = blockIdx.x;
    const uint32_t total_wg = KVS ? p.total * p.ksplit : p.total;      // (KVS: the splits of a query tile are neighbours, so the K / V of a (batch, head) still meet in one L2)
    if ((total_wg & 7u) == 0) logical = (blockIdx.x & 7u) * (total_wg >> 3) + (blockIdx.x >> 3);
    uint32_t ks = 0;
    if constexpr (KVS) { const uint32_t lq = logical / p.ksplit; ks = logical - lq * p.ksplit; logical = lq; }
    const uint32_t bh = logical / p.nqt, qt = logical - bh * p.nqt;
    const uint32_t b = bh / (uint32_t)p.H, hh = bh - b * (uint32_t)p.H;

    const uint32_t q0 = qt * QWG + wave * QW + c32;               // this lane's query of block qb is q0 + 32 qb
    const T* Qb = reinterpret_cast<const T*>(p.q) + (int64_t)b * p.Sq * p.ldq + hh * (uint32_t)D;
    const T* K = reinterpret_cast<const T*>(p.k) + (int64_t)b * p.Sk * p.ldk + hh * (uint32_t)D;
    const T* VT = reinterpret_cast<const T*>(p.vt) + ((int64_t)b * p.H + hh) * D * (int64_t)p.Sk;

    if (DU & 1 || D & 31) {      // zero the LDS once: padding units of K rows / padding rows of V^T are never written by the staging
        for (uint32_t i = tid; i < NSTG * STAGE / 16; i += 256) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if constexpr (LM) {      // V^T row D of every stage: ones for all 64 keys
            const T one = (T)1.0f;
            if (tid < NSTG * 64u) *reinterpret_cast<T*>(lds + (tid >> 6) * STAGE + 64u * KP + (uint32_t)D * VP + (tid & 63u) * 2u) = one;
        }
        __syncthreads();
    }

    v8 qreg[QB][KC];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            u32x4 w = u32x4{0u, 0u, 0u, 0u};
            const uint32_t qi = q0 + 32u * qb;
            const T* Q = Qb + (int64_t)((!RAGGED || qi < (uint32_t)p.Sq) ? qi : 0u) * p.ldq;
            if (2 * kc + 1 < DU || h == 0) w = *reinterpret_cast<const u32x4*>(Q + kc * 16 + h * 8);   // unit 2kc+h < DU
            // Pass Q through a VALU move before the loop: the loop's MFMAs then read ALU results, not load results, so the waitcnt pass
            // does not place `s_waitcnt vmcnt` (which would drain the in-flight K / V prefetch) in front of them.
#pragma unroll
            for (int j = 0; j < 4; ++j) { uint32_t e = w[j]; asm volatile("v_mov_b32 %0, %0" : "+v"(e)); w[j] = e; }
            qreg[qb][kc] = __builtin_bit_cast(v8, w);
        }

    // staging roles: 64*DU 16-byte units per tile and operand.  K unit u: key row u / DU, column u % DU; V^T unit u: row u / 8 (< D),
    // key column u % 8.  (Named scalars + macros: arrays or lambdas holding the prefetch registers end up in scratch.)
    // NU <= 5; unit i of this thread is u = tid + 256 i
    constexpr bool ALLV = NU * 256 == 64 * DU;                 // every unit slot of every thread is a real unit
#define GSW_ATTN_UNIT_DECL(i)                                                                           \
    uint4 kreg##i = make_uint4(0, 0, 0, 0), vreg##i = make_uint4(0, 0, 0, 0);                           \
    const uint32_t u##i = tid + 256u * i;                                                               \
    const bool uv##i = u##i < 64u * DU;                                                                 \
    const uint32_t uu##i = uv##i ? u##i : 0u;                                                           \
    const uint32_t kr##i = uu##i / (uint32_t)DU, kcol##i = uu##i - kr##i * (uint32_t)DU;                \
    const T* kg##i = K + (int64_t)kr##i * p.ldk + kcol##i * 8u;                                         \
    const uint32_t kst##i = kr##i * KP + kcol##i * 16u;                                                 \
    const T* vg##i = VT + (int64_t)(uu##i >> 3) * p.Sk + (uu##i & 7u) * 8u;                             \
    const uint32_t vst##i = 64u * KP + (uu##i >> 3) * VP + ((uu##i & 7u) >> 1) * 32u + (uu##i & 1u) * 8u;      /* unit j (keys 8j..8j+7) of 16-key block j >> 1: groups at slots (j & 1), (j & 1) + 2 */
    GSW_ATTN_UNIT_DECL(0)
    GSW_ATTN_UNIT_DECL(1)
    GSW_ATTN_UNIT_DECL(2)
    GSW_ATTN_UNIT_DECL(3)
    GSW_ATTN_UNIT_DECL(4)
#define GSW_ATTN_GLOAD1(i, key0)                                                                        \
    if (NU > i && (ALLV || uv##i)) {                                                                    \
        if (RAGGED) {                                                                                   \
            const int32_t row_ = min((int32_t)(key0) + (int32_t)kr##i, p.Sk - 1);                       \
            const int32_t col_ = (int32_t)(key0) + (int32_t)(uu##i & 7u) * 8;                           \
            kreg##i = *reinterpret_cast<const uint4*>(K + (int64_t)row_ * p.ldk + kcol##i * 8u);        \
            vreg##i = *reinterpret_cast<const uint4*>(VT + (int64_t)(uu##i >> 3) * p.Sk + (col_ + 8 <= p.Sk ? col_ : 0)); \
        } else {                                                                                        \
            kreg##i = *reinterpret_cast<const uint4*>(kg##i + (int64_t)(key0) * p.ldk);                 \
            vreg##i = *reinterpret_cast<const uint4*>(vg##i + (key0));                                  \
        }                                                                                               \
    }
#define GSW_ATTN_GLOAD(key0) GSW_ATTN_GLOAD1(0, key0) GSW_ATTN_GLOAD1(1, key0) GSW_ATTN_GLOAD1(2, key0) GSW_ATTN_GLOAD1(3, key0) GSW_ATTN_GLOAD1(4, key0)
#define GSW_ATTN_LSTORE1(i, st)                                                                         \
    if (NU > i && (ALLV || uv##i)) {                                                                    \
        uint8_t* base_ = lds + (st) * STAGE;                                                            \
        *reinterpret_cast<uint4*>(base_ + kst##i) = kreg##i;                                            \
        *reinterpret_cast<uint2*>(base_ + vst##i) = make_uint2(vreg##i.x, vreg##i.y);       /* keys 8j .. 8j+3   -> slot (j & 1)     */ \
        *reinterpret_cast<uint2*>(base_ + vst##i + 16u) = make_uint2(vreg##i.z, vreg##i.w); /* keys 8j+4 .. 8j+7 -> slot (j & 1) + 2 */ \
    }
#define GSW_ATTN_LSTORE(st) GSW_ATTN_LSTORE1(0, st) GSW_ATTN_LSTORE1(1, st) GSW_ATTN_LSTORE1(2, st) GSW_ATTN_LSTORE1(3, st) GSW_ATTN_LSTORE1(4, st)
    static_assert(NU <= 5, "staging code covers up to 5 units per thread");

    f32x16 o[QB][DB];
    f32x4 ot[QB][2];          // TAIL: rows DBF*32 + 4 (lane >> 4) + j of queries 0-15 | 16-31 of the block (query = lane & 15)
    float m_i[QB], l_i[QB];
    f32x4 l4[QB][2];          // L4: two accumulator chains per query block; all four elements of an accumulator are the same number
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        ot[qb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; ot[qb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qb][db][i] = 0.f;
        m_i[qb] = -INFINITY;
        l_i[qb] = 0.f;
        l4[qb][0] = f32x4{0.f, 0.f, 0.f, 0.f}; l4[qb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float cs = p.scale_log2;

    const int32_t nt = (p.Sk + 63) >> 6;
    // one 64-key tile: S^T, online softmax, O^T accumulation
    // DMA: the lane's eight possible chunk offsets inside its fragment row (row c32 of a 32-row block; chunk k of the row sits at slot k ^ ((c32 >> 1) & 7))
    uint32_t swo[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) swo[k] = DMA ? c32 * 128u + (((uint32_t)k ^ ((c32 >> 1) & 7u)) << 4) : 0u;
    auto tile = [&](const uint8_t* Kl, const uint8_t* Vl, int32_t t) __attribute__((always_inline)) {

        // ---- S^T = K Q^T : two 32-key blocks per query block; 2 * QB independent accumulator chains, interleaved
        f32x16 s[QB][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int i = 0; i < 16; ++i) { s[qb][0][i] = 0.f; s[qb][1][i] = 0.f; }
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const v8 a = DMA ? *reinterpret_cast<const v8*>(Kl + (uint32_t)(kb * 32 * 128) + (h ? swo[(2 * kc + 1) & 7] : swo[(2 * kc) & 7]))
                                 : *reinterpret_cast<const v8*>(Kl + (uint32_t)(kb * 32 + (int)c32) * KP + (uint32_t)kc * 32u + h * 16u);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb][kb] = AT<T>::mfma(a, qreg[qb][kc], s[qb][kb]);
            }
        }
        if ((t + 1) * 64 > p.Sk_valid) {            // padded keys of the last tile(s) (cross-attention, 77 context tokens): score -inf
            asm volatile("; masked tile");             // (a wave-uniform branch the steady state jumps over; the comment marks the block for tools/isa_loop_mix.py)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int32_t key = DMA ? t * 64 + kb * 32 + (i >> 3) * 16 + (int32_t)h * 8 + ((i >> 2) & 1) * 4 + (i & 3)      // (the K rows of a DMA tile are permuted)
                                            : t * 64 + kb * 32 + (i >> 2) * 8 + (int32_t)h * 4 + (i & 3);
                    if (key >= p.Sk_valid) {
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) s[qb][kb][i] = -INFINITY;
                    }
                }
        }

        // ---- online softmax (base-2 domain); a query lives in lanes c32 and c32 + 32
        v8 pb[QB][2][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float mx = s[qb][0][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[qb][0][i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[qb][1][i]);
            mx = xhalf_max(mx);
            const float m_new = fmaxf(m_i[qb], mx * cs);
            const float alpha = __builtin_amdgcn_exp2f(m_i[qb] - m_new);      // raw v_exp_f32: arguments are <= 0, underflow to 0 is the intent
            m_i[qb] = m_new;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(s[qb][kb][i], cs, -m_new));
                    pb[qb][kb][i >> 3][i & 7] = (T)e;
                }
            }
            if (__any(alpha != 1.0f)) {                 // once the running maxima have settled the whole wave skips the rescale
                if constexpr (L4) { l4[qb][0] *= alpha; l4[qb][1] *= alpha; }
#pragma unroll
                for (int db = 0; db < DBF; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[qb][db][i] *= alpha;
                if constexpr (TAIL) {
                    float a0, a1;
                    rows16_split(alpha, a0, a1);
                    ot[qb][0] *= a0; ot[qb][1] *= a1;
                }
            }
        }

        // ---- O^T += V^T P^T
        if constexpr (L4) {
            const v4 ones = v4{(T)1.0f, (T)1.0f, (T)1.0f, (T)1.0f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        const v8 pv = pb[qb][kb][tt];
                        l4[qb][0] = AT<T>::mfma4(ones, v4{pv[0], pv[1], pv[2], pv[3]}, l4[qb][0]);
                        l4[qb][1] = AT<T>::mfma4(ones, v4{pv[4], pv[5], pv[6], pv[7]}, l4[qb][1]);
                    }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int db = 0; db < DBF; ++db) {
                    const v8 a = DMA ? *reinterpret_cast<const v8*>(Vl + (uint32_t)(db * 32 * 128) + (h ? swo[(4 * kb + 2 * tt + 1) & 7] : swo[(4 * kb + 2 * tt) & 7]))
                                     : *reinterpret_cast<const v8*>(Vl + (uint32_t)(db * 32 + (int)c32) * VP + (uint32_t)(kb * 32 + tt * 16) * 2u + h * 16u);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) o[qb][db] = AT<T>::mfma(a, pb[qb][kb][tt], o[qb][db]);
                }
            }
            if constexpr (TAIL) {
                // rows DBF*32 .. + 15 against all 32 keys of the half tile: one 16 x 16 x 32 MFMA per 16 queries.  Lane (row r = lane >> 4, m = lane & 15): V^T row
                // DBF*32 + m, keys (r & 1) * 16 + (r >> 1) * 4 + {0..3, 8..11} -- the key order the swapped probabilities carry
                const uint32_t r16 = lane >> 4, m16 = lane & 15u;
                const v8 a = *reinterpret_cast<const v8*>(Vl + (uint32_t)(DBF * 32 + (int)m16) * VP + (uint32_t)(kb * 32 + (int)((r16 & 1u) * 16u)) * 2u + (r16 >> 1) * 16u);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    u32x4 b0 = __builtin_bit_cast(u32x4, pb[qb][kb][0]), b1 = __builtin_bit_cast(u32x4, pb[qb][kb][1]);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(b0[w], b1[w], false, false);
                        b0[w] = sw[0]; b1[w] = sw[1];
                    }
                    ot[qb][0] = AT<T>::mfma16(a, __builtin_bit_cast(v8, b0), ot[qb][0]);      // queries 0-15 of the block
                    ot[qb][1] = AT<T>::mfma16(a, __builtin_bit_cast(v8, b1), ot[qb][1]);      // queries 16-31
                }
            }
        }
    };
    if constexpr (DMA) {
        __shared__ __attribute__((aligned(16))) uint8_t sg0[16384], sg1[16384], sg2[16384], sg3[16384];      // a stage: 64 K rows, then 64 V^T rows, 128 bytes each
        // this wave's pieces of a tile: K pieces wave, wave + 4 and V^T pieces wave, wave + 4 (a piece = eight LDS rows = one instruction).  Lane l of piece pc: LDS row
        // 8 pc + (l >> 3), slot l & 7 <- chunk (l & 7) ^ ((row >> 1) & 7) of source row: key pi(row) of the tile (K) / head row `row` of V^T
        const T* kp[2];
        const T* vp[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t pc = wave + 4u * j, row = 8u * pc + (lane >> 3);
            const uint32_t chunk = (lane & 7u) ^ ((row >> 1) & 7u);
            const uint32_t key = 16u * (pc >> 1) + 8u * (lane >> 5) + 4u * (pc & 1u) + ((lane >> 3) & 3u);
            kp[j] = K + (int64_t)key * p.ldk + chunk * 8u;
            vp[j] = VT + (int64_t)row * p.Sk + chunk * 8u;
        }
        const uint32_t wv = __builtin_amdgcn_readfirstlane(wave);
        const int64_t kstep = (int64_t)64 * p.ldk;
        auto dma = [&](uint8_t* sg, int32_t t) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kp[j] + (int64_t)t * kstep),
                                                 (__attribute__((address_space(3))) void*)(sg + (wv + 4u * j) * 1024u), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vp[j] + (int64_t)t * 64),
                                                 (__attribute__((address_space(3))) void*)(sg + 8192u + (wv + 4u * j) * 1024u), 16, 0, 0);
            }
        };
        dma(sg0, 0);
        dma(sg1, 1);
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), as a builtin: hipcc's waitcnt pass counts the DMA itself
        __syncthreads();
        for (int32_t t = 0; t < nt; t += 4) {          // nt % 4 == 0 (host-checked)
            // tiles t, t + 1 are visible in stages 0, 1; t + 2, t + 3 land in stages 2, 3 under them; ONE barrier per two tiles
            dma(sg2, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            tile(sg0, sg0 + 8192, t);
            dma(sg3, t + 3);
            __builtin_amdgcn_sched_barrier(0);
            tile(sg1, sg1 + 8192, t + 1);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
            const int32_t tn = t + 4 < nt ? t + 4 : t;      // (the last round re-fetches its own tiles into the idle stages: no conditional DMA)
            dma(sg0, tn);
            __builtin_amdgcn_sched_barrier(0);
            tile(sg2, sg2 + 8192, t + 2);
            dma(sg1, tn + 1);
            __builtin_amdgcn_sched_barrier(0);
            tile(sg3, sg3 + 8192, t + 3);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
    } else if (PAIR) {
        GSW_ATTN_GLOAD(0)
        GSW_ATTN_LSTORE(0u)
        GSW_ATTN_GLOAD((nt > 1 ? 1 : 0) << 6)
        GSW_ATTN_LSTORE(1u)
        __syncthreads();
        for (int32_t t = 0; t < nt; t += 2) {
            // tiles t and t+1 are visible; t+2 and t+3 are fetched, written into the two idle stages, and published by ONE barrier
            GSW_ATTN_GLOAD((t + 2 < nt ? t + 2 : nt - 1) << 6)
            __builtin_amdgcn_sched_barrier(0);
            tile(lds + (uint32_t)(t & 3) * STAGE, lds + (uint32_t)(t & 3) * STAGE + 64u * KP, t);
            GSW_ATTN_LSTORE((uint32_t)((t + 2) & 3))
            if (t + 1 < nt) {
                GSW_ATTN_GLOAD((t + 3 < nt ? t + 3 : nt - 1) << 6)
                __builtin_amdgcn_sched_barrier(0);
                tile(lds + (uint32_t)((t + 1) & 3) * STAGE, lds + (uint32_t)((t + 1) & 3) * STAGE + 64u * KP, t + 1);
                GSW_ATTN_LSTORE((uint32_t)((t + 3) & 3))
            }
            __syncthreads();
        }
    } else {
        // (KVS: this workgroup's share of the key tiles)
        const int32_t t_lo = KVS ? (int32_t)(((int64_t)ks * nt) / (int32_t)p.ksplit) : 0;
        const int32_t t_hi = KVS ? (int32_t)(((int64_t)(ks + 1u) * nt) / (int32_t)p.ksplit) : nt;
        GSW_ATTN_GLOAD(t_lo << 6)
        GSW_ATTN_LSTORE(0u)
        __syncthreads();
        for (int32_t t = t_lo; t < t_hi; ++t) {
            // prefetch the next tile into registers -- unconditionally (the last iteration re-fetches its own tile into the idle stage):
            // a conditional load made the compiler merge the registers right after the branch, i.e. wait for HBM inside the MFMA phase
            const int32_t tn = t + 1 < t_hi ? t + 1 : t;
            GSW_ATTN_GLOAD(tn << 6)
            __builtin_amdgcn_sched_barrier(0);          // keep the prefetch at the top of the iteration (the scheduler sinks it otherwise)
            tile(lds + (uint32_t)((t - t_lo) & 1) * STAGE, lds + (uint32_t)((t - t_lo) & 1) * STAGE + 64u * KP, t);
            GSW_ATTN_LSTORE((uint32_t)((t - t_lo + 1) & 1))
            __syncthreads();
        }
    }

    if constexpr (KVS) {
        // ---- partial result of this key range: accumulators as they are (one coalesced 256-byte store per register), then m and the lane's partial row sum
        float* base = p.part + (((size_t)logical * p.ksplit + ks) * 4u + wave) * (size_t)((NACC + 2) * 64) + lane;
#pragma unroll
        for (int db = 0; db < DBF; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) base[(db * 16 + i) * 64] = o[0][db][i];
        if constexpr (TAIL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) base[(DBF * 16 + i) * 64] = ot[0][i >> 2][i & 3];
        }
        base[NACC * 64] = m_i[0];
        base[(NACC + 1) * 64] = l4[0][0][0] + l4[0][1][0];      // (KVS kernels are L4)
        return;
    }

    // ---- normalise and store: lane holds O^T[d][query c32] for d = db*32 + (i/4)*8 + h*4 + (i%4)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float l;
        if constexpr (LM) {      // row 40 of the tail: element 0 of lanes 32-47 (lane row 2), query 16 hq + (lane & 15)
            const int idx = (int)(4u * (32u + (lane & 15u)));
            const float l0 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(ot[qb][0][0])));
            const float l1 = __int_as_float(__builtin_amdgcn_ds_bpermute(idx, __float_as_int(ot[qb][1][0])));
            l = (c32 & 16u) ? l1 : l0;
            l_i[qb] = 0.5f * l;      // (the tail's normalisation below adds the two lane halves again)
        } else {
            if constexpr (L4) l_i[qb] = l4[qb][0][0] + l4[qb][1][0];
            l = xhalf_sum(l_i[qb]);
        }
        const float inv = 1.0f / l;
        const uint32_t qi = q0 + 32u * qb;
        if (RAGGED && qi >= (uint32_t)p.Sq) continue;
        T* O = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Sq + qi) * p.ldo + hh * (uint32_t)D;
#pragma unroll
        for (int db = 0; db < DBF; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = db * 32 + g * 8 + (int)h * 4;
                v4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(o[qb][db][g * 4 + j] * inv);
                *reinterpret_cast<v4*>(O + d0) = w;
            }
        }
    }
    if constexpr (TAIL) {
        // the 16-row tail: lane (r = lane >> 4, n = lane & 15) holds rows DBF*32 + 4 r + j of query n (first MFMA) / 16 + n (second) of each 32-query block
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float inv2[2];
            rows16_split(1.0f / xhalf_sum(l_i[qb]), inv2[0], inv2[1]);
            const int d0 = DBF * 32 + (int)(lane >> 4) * 4;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const uint32_t qi = qt * QWG + wave * QW + 32u * qb + 16u * hq + (lane & 15u);
                if ((RAGGED && qi >= (uint32_t)p.Sq) || d0 >= D) continue;
                T* O = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Sq + qi) * p.ldo + hh * (uint32_t)D;
                v4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = (T)(ot[qb][hq][j] * inv2[hq]);
                *reinterpret_cast<v4*>(O + d0) = w;
            }
        }
    }
#undef GSW_ATTN_GLOAD
#undef GSW_ATTN_LSTORE
#undef GSW_ATTN_GLOAD1
#undef GSW_ATTN_LSTORE1
#undef GSW_ATTN_UNIT_DECL
}

// Second half of a key-split attention launch: one workgroup per query tile with the thread geometry of the first half; every lane merges its `ksplit` partial
// accumulators (fixed order: deterministic), normalises and stores like the unsplit kernel.
template <typename T, int DU>
__global__ __launch_bounds__(256) void gsw_attn_combine_kernel(AttnArgs p) {
    using v4 = typename AT<T>::v4;
    constexpr int D = DU * 8, DBF = D / 32, TR = D % 32, DB = DBF > 0 ? DBF : 1, NACC = DBF * 16 + (TR ? 8 : 0);
    constexpr bool TAIL = TR != 0;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, h = lane >> 5, c32 = lane & 31u;
    const uint32_t logical = blockIdx.x;
    const uint32_t bh = logical / p.nqt, qt = logical - bh * p.nqt;
    const uint32_t b = bh / (uint32_t)p.H, hh = bh - b * (uint32_t)p.H;
    const uint32_t qi = qt * 128u + wave * 32u + c32;
    constexpr size_t REC = (size_t)(NACC + 2) * 64;
    const float* base = p.part + ((size_t)logical * p.ksplit * 4u + wave) * REC + lane;
    float m = -INFINITY;
    for (uint32_t s = 0; s < p.ksplit; ++s) m = fmaxf(m, base[(size_t)s * 4u * REC + NACC * 64]);
    f32x16 o[DB];
    f32x4 ot[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[db][i] = 0.f;
    float l = 0.f;
    for (uint32_t s = 0; s < p.ksplit; ++s) {
        const float* r = base + (size_t)s * 4u * REC;
        const float sc = __builtin_amdgcn_exp2f(r[NACC * 64] - m);          // (m and the row sums are per query of the 32-lane-half layout)
        l = fmaf(r[(NACC + 1) * 64], sc, l);
#pragma unroll
        for (int db = 0; db < DBF; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[db][i] = fmaf(r[(db * 16 + i) * 64], sc, o[db][i]);
        if constexpr (TAIL) {
            float s2[2];
            rows16_split(sc, s2[0], s2[1]);                                  // the tail accumulators keep one query per lane of a 16-lane row
#pragma unroll
            for (int i = 0; i < 8; ++i) ot[i >> 2][i & 3] = fmaf(r[(DBF * 16 + i) * 64], s2[i >> 2], ot[i >> 2][i & 3]);
        }
    }
    const float inv = 1.0f / xhalf_sum(l);
    T* O = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Sq + qi) * p.ldo + hh * (uint32_t)D;
#pragma unroll
    for (int db = 0; db < DBF; ++db) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = db * 32 + g * 8 + (int)h * 4;
            v4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(o[db][g * 4 + j] * inv);
            *reinterpret_cast<v4*>(O + d0) = w;
        }
    }
    if constexpr (TAIL) {
        float inv2[2];
        rows16_split(inv, inv2[0], inv2[1]);
        const int d0 = DBF * 32 + (int)(lane >> 4) * 4;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
            const uint32_t qh = qt * 128u + wave * 32u + 16u * hq + (lane & 15u);
            if (d0 >= D) continue;
            T* Oh = reinterpret_cast<T*>(p.o) + ((int64_t)b * p.Sq + qh) * p.ldo + hh * (uint32_t)D;
            v4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (T)(ot[hq][j] * inv2[hq]);
            *reinterpret_cast<v4*>(Oh + d0) = w;
        }
    }
}

}  // namespace

template <typename T, int QB, int DU>
static int launch_attn_cfg(const AttnArgs& a, uint32_t grid, bool ragged, float* ws, int64_t ws_bytes, hipStream_t st) {
    constexpr int KC = (DU + 1) / 2, DBF = DU * 8 / 32, TR = DU * 8 % 32, NACC = DBF * 16 + (TR ? 8 : 0);
    constexpr uint32_t stage = 64 * (KC * 32 + 16) + (DBF * 32 + (TR ? 16 : 0)) * VP;
    // Key-split form: few query tiles against many key tiles (one image's self-attention at 64 x 64: 62 -> 49 us).  As many splits as keep every workgroup
    // resident at once (2 per CU); only from 32 key tiles up -- the second launch and the partial results cost ~12 us, and at 16 tiles (32 x 32: 17.6 -> 21.4 us)
    // that is more than the split saves.  Needs the caller's workspace (gsw_attention_ws); GSW_ATTN_KVS=0 switches it off (A/B), k > 1 forces k ways from 8 tiles up.
    if constexpr (QB == 1 && (DU == 8 || DU == 5)) {
        static const int kvs_env = getenv("GSW_ATTN_KVS") ? atoi(getenv("GSW_ATTN_KVS")) : 1;
        const int32_t nt = a.Sk >> 6;
        if (!ragged && kvs_env && ws && a.Sk_valid == a.Sk && a.total <= 256u && nt >= (kvs_env > 1 ? 8 : 32)) {
            uint32_t ksplit = (uint32_t)std::min<int64_t>(std::min<int64_t>(512 / a.total, nt / 4), 8);
            if (kvs_env > 1) ksplit = (uint32_t)std::min<int>(kvs_env, nt);                                   // (tests: force a split count)
            const int64_t need = (int64_t)a.total * ksplit * 4 * (NACC + 2) * 64 * (int64_t)sizeof(float);
            if (ksplit >= 2 && need <= ws_bytes) {
                AttnArgs k = a;
                k.ksplit = ksplit; k.part = ws;
                hipLaunchKernelGGL((gsw_attn_fwd_kernel<T, 1, DU, false, false, true>), dim3(a.total * ksplit), dim3(256), 2 * stage, st, k);
                hipLaunchKernelGGL((gsw_attn_combine_kernel<T, DU>), dim3(a.total), dim3(256), 0, st, k);
                return GSW_OK;
            }
        }
    }
    static const int pair_env = getenv("GSW_ATTN_PAIR") ? atoi(getenv("GSW_ATTN_PAIR")) : 1;      // A/B switch for profiling
    const bool pair = !ragged && pair_env && 4 * stage <= 80 * 1024 && (a.Sk >> 6) >= 16;      // long key sequences only: +2 % at 4096 keys, a loss at 256
    const uint32_t lds = (pair ? 4 : 2) * stage;
    const void* fn = pair ? (const void*)gsw_attn_fwd_kernel<T, QB, DU, true, false>
                          : (ragged ? (const void*)gsw_attn_fwd_kernel<T, 1, DU, false, true> : (const void*)gsw_attn_fwd_kernel<T, QB, DU, false, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    }
    if constexpr (DU == 8) {
        static const int dma_env = getenv("GSW_ATTN_DMA") ? atoi(getenv("GSW_ATTN_DMA")) : 1;      // A/B switch
        if (pair && dma_env && ((a.Sk >> 6) & 3) == 0) {      // K / V^T tiles by LDS-DMA into four static stages (64 KiB)
            hipLaunchKernelGGL((gsw_attn_fwd_kernel<T, QB, DU, true, false, false, true>), dim3(grid), dim3(256), 0, st, a);
            return GSW_OK;
        }
    }
    if (pair) hipLaunchKernelGGL((gsw_attn_fwd_kernel<T, QB, DU, true, false>), dim3(grid), dim3(256), lds, st, a);
    else if (ragged) hipLaunchKernelGGL((gsw_attn_fwd_kernel<T, 1, DU, false, true>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((gsw_attn_fwd_kernel<T, QB, DU, false, false>), dim3(grid), dim3(256), lds, st, a);
    return GSW_OK;
}

template <typename T>
static int launch_attn(const AttnArgs& a, int head_dim, int QB, uint32_t grid, bool ragged, float* ws, int64_t wsb, hipStream_t st) {
    if (head_dim == 64) return QB == 2 ? launch_attn_cfg<T, 2, 8>(a, grid, ragged, ws, wsb, st) : launch_attn_cfg<T, 1, 8>(a, grid, ragged, ws, wsb, st);
    if (head_dim == 40) return QB == 2 ? launch_attn_cfg<T, 2, 5>(a, grid, ragged, ws, wsb, st) : launch_attn_cfg<T, 1, 5>(a, grid, ragged, ws, wsb, st);
    if (head_dim == 80) return launch_attn_cfg<T, 1, 10>(a, grid, ragged, ws, wsb, st);
    return launch_attn_cfg<T, 1, 20>(a, grid, ragged, ws, wsb, st);
}

int gsw_attention(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev, int B, int H, int head_dim, int Sq, int Sk, int Sk_valid,
                  int ldq, int ldk, int ldo, float scale, int dtype, void* stream) {
    return gsw_attention_ws(q_dev, k_dev, vt_dev, out_dev, B, H, head_dim, Sq, Sk, Sk_valid, ldq, ldk, ldo, scale, dtype, nullptr, 0, stream);
}

int gsw_attention_ws(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev, int B, int H, int head_dim, int Sq, int Sk, int Sk_valid,
                     int ldq, int ldk, int ldo, float scale, int dtype, void* workspace_dev, int64_t workspace_bytes, void* stream) {
    if (workspace_bytes < 0 || (workspace_bytes > 0 && !workspace_dev) || ((uintptr_t)workspace_dev & 15)) return GSW_ERR_BAD_ARG;
    // q: [B, Sq, >= H*head_dim] (row stride ldq), k: [B, Sk, >= H*head_dim] (row stride ldk), vt: [B, H*head_dim, Sk] contiguous (V
    // transposed), out: [B, Sq, >= H*head_dim] (row stride ldo).  head_dim 40 / 64 / 80 / 160; any Sq; Sk % 8 == 0; row strides
    // multiples of 8 elements; keys in [Sk_valid, Sk) are padding and get zero weight.
    if (!q_dev || !k_dev || !vt_dev || !out_dev || B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0 || Sk_valid <= 0 || Sk_valid > Sk) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (head_dim != 40 && head_dim != 64 && head_dim != 80 && head_dim != 160) return GSW_ERR_UNSUPPORTED;
    const int inner = H * head_dim;
    if ((Sk & 7) || ldq < inner || ldk < inner || ldo < inner || ((ldq | ldk | ldo) & 7)) return GSW_ERR_UNSUPPORTED;
    static const int qb_env = getenv("GSW_ATTN_QB") ? atoi(getenv("GSW_ATTN_QB")) : 2;      // A/B switch for profiling
    const bool ragged = (Sq & 127) || (Sk & 63);
    // 256-query workgroups once there are plenty of them (7-12 % faster from ~1000 workgroups up); below ~400 of them the 128-query form keeps more CUs
    // busy (one image at 64 x 64: 62 vs 91 us; profiles/r03y_attention_query_tile_sweep.txt)
    const int64_t wg256 = (int64_t)(Sq / 256) * B * H;
    const int QB = ((Sq & 255) == 0 && Sq >= 512 && qb_env >= 2 && head_dim < 80 && !ragged && (wg256 > 400 || qb_env == 3)) ? 2 : 1;      // (GSW_ATTN_QB: 1 / 3 force a form)
    const int64_t total = (int64_t)((Sq + 128 * QB - 1) / (128 * QB)) * B * H;
    if (total > 0x7FFFFFFF) return GSW_ERR_UNSUPPORTED;
    AttnArgs a;
    a.q = q_dev; a.k = k_dev; a.vt = vt_dev; a.o = out_dev;
    a.H = H; a.Sq = Sq; a.Sk = Sk; a.ldq = ldq; a.ldk = ldk; a.ldo = ldo; a.Sk_valid = Sk_valid;
    a.scale_log2 = scale * 1.4426950408889634f;
    a.nqt = (uint32_t)((Sq + 128 * QB - 1) / (128 * QB));
    a.total = (uint32_t)total;
    a.ksplit = 1; a.part = nullptr;
    float* const ws = workspace_bytes > 0 ? (float*)workspace_dev : nullptr;
    const int rc = dtype == GSW_F16 ? launch_attn<_Float16>(a, head_dim, QB, (uint32_t)total, ragged, ws, workspace_bytes, (hipStream_t)stream)
                                    : launch_attn<__bf16>(a, head_dim, QB, (uint32_t)total, ragged, ws, workspace_bytes, (hipStream_t)stream);
    if (rc != GSW_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Row softmax in place: x[r, 0:cols] <- softmax(scale * x[r, 0:cols]).  The single-head, 512-wide attention of the VAE mid block
// (diffusers AutoencoderKL, reached from extract.py:41 `vae.encode` and the pipelines' decode_image) does not fit the flash kernel's
// register budget, so it runs as two products on the matmul engine (S = Q K^T per image, O = P V) with this kernel in between.
// One workgroup per row, three passes over the row (max, sum of exponentials, write); the row (<= 18 KiB at 768x768) stays in L2.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gsw_softmax_rows_kernel(T* __restrict__ x, int64_t ld, int32_t cols, float scale_log2) {
    __shared__ float red[4];
    T* row = x + (int64_t)blockIdx.x * ld;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const int32_t nv = cols >> 3;
    auto load8 = [&](int32_t v, float (&f)[8]) {
        const uint4 u = reinterpret_cast<const uint4*>(row)[v];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            T lo, hi;
            const uint16_t a = (uint16_t)w[k], b = (uint16_t)(w[k] >> 16);
            __builtin_memcpy(&lo, &a, 2); __builtin_memcpy(&hi, &b, 2);
            f[2 * k] = (float)lo; f[2 * k + 1] = (float)hi;
        }
    };
    auto wg_reduce = [&](float v, bool is_max) -> float {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o); v = is_max ? fmaxf(v, t) : v + t; }
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const float a = red[0], b = red[1], c = red[2], d = red[3];
        return is_max ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
    };
    float mx = -INFINITY;
    for (int32_t v = tid; v < nv; v += 256) {
        float f[8];
        load8(v, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, f[k]);
    }
    mx = wg_reduce(mx, true) * scale_log2;
    float sum = 0.f;
    for (int32_t v = tid; v < nv; v += 256) {
        float f[8];
        load8(v, f);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += __builtin_amdgcn_exp2f(fmaf(f[k], scale_log2, -mx));
    }
    const float inv = 1.0f / wg_reduce(sum, false);
    for (int32_t v = tid; v < nv; v += 256) {
        float f[8];
        load8(v, f);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T lo = (T)(__builtin_amdgcn_exp2f(fmaf(f[2 * k], scale_log2, -mx)) * inv), hi = (T)(__builtin_amdgcn_exp2f(fmaf(f[2 * k + 1], scale_log2, -mx)) * inv);
            uint16_t a, b;
            __builtin_memcpy(&a, &lo, 2); __builtin_memcpy(&b, &hi, 2);
            w[k] = (uint32_t)a | ((uint32_t)b << 16);
        }
        reinterpret_cast<uint4*>(row)[v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int gsw_softmax_rows(void* x_dev, int64_t rows, int cols, int64_t ld, float scale, int dtype, void* stream) {
    if (!x_dev || rows < 0 || cols <= 0 || ld < cols) return GSW_ERR_BAD_ARG;
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if ((cols & 7) || (ld & 7) || rows > 0x7FFFFFFF) return GSW_ERR_UNSUPPORTED;
    if (rows == 0) return GSW_OK;
    const float sl2 = scale * 1.4426950408889634f;
    if (dtype == GSW_F16) hipLaunchKernelGGL(gsw_softmax_rows_kernel<_Float16>, dim3((uint32_t)rows), dim3(256), 0, (hipStream_t)stream, (_Float16*)x_dev, ld, cols, sl2);
    else hipLaunchKernelGGL(gsw_softmax_rows_kernel<__bf16>, dim3((uint32_t)rows), dim3(256), 0, (hipStream_t)stream, (__bf16*)x_dev, ld, cols, sl2);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

