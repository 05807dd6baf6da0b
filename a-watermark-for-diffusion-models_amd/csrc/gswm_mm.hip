// gswm_mm.hip -- the matmul engine of the eps model (rows X2 / G1 of SURVEY.md section 8a): every transformer linear of the UNet
// (diffusers BasicTransformerBlock / Transformer2DModel behind extract.py:66-69) and, through a tap table, the PF convolutions.
// gfx950 only.
//
//   Y[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n] + rowbias[image(m), n] + resid[m, n]),   fp16 / bf16 in, fp32 accumulate
//
// Structure (one persistent workgroup per CU, grid = 256, walking output tiles in an XCD-aware panel order):
//   * Output tile BM x BN = 256 x 160 (MT = 4) or 128 x 160 (MT = 2, for launches whose 256-row tiling would leave CUs idle).  8 multiplying
//     waves in LOCKSTEP as 4 (M) x 2 (N); a wave owns 64 (32) x 80 outputs as MT x 5 v_mfma_f32_16x16x32 accumulators.
//   * K is consumed in STAGES of 64 k-values = two PHASES of 32.  A stage (BM + BN rows x 128 B = 52 KiB) goes global -> LDS by DMA
//     (global_load_lds, 16 B per lane, 1 KiB per wave instruction) into a THREE-stage ring (156 of the 160 KiB).  Waits are counted
//     (s_waitcnt vmcnt(N): the newest stage stays in flight across the barrier), barriers are raw s_barrier, ONE per stage.
//   * Who issues the DMA: the 8-wave variant interleaves a stage's pieces, in two halves, sparsely between the MFMAs of two phases
//     (sched_group_barrier pins the order); the 12-wave variant (SPLIT) adds four producer waves (one per SIMD) that issue all 52 pieces,
//     so the multiplying waves carry no DMA and no tap bookkeeping (+7-9 % on the 3x3 convolutions).
//   * LDS rows are 128 B (64 k-values); 16-byte chunks are XOR-swizzled with (row >> 1) & 7 on the DMA SOURCE address, which makes
//     every ds_read_b128 fragment read conflict-free without padding.
//   * A-operand rows come from up to three K SEGMENTS (pointer, row stride, channel count, tap table): a dense matrix is one segment
//     with one tap; a 3x3 convolution on a padded-flat NHWC tensor is one segment with nine taps (a tap = a constant row offset, see
//     gswm_conv.hip) whose M dimension enumerates INTERIOR pixels only; the resnet's conv2 + 1x1 shortcut of cat(x, skip) is three
//     segments.  The producer only adds uniform steps to per-lane pointers per stage; tap / channel-block / segment / tile bookkeeping
//     runs once per RUN of stages in a scalar slow path.
//   * Epilogues straight from the accumulators (no LDS image, no barrier): bias, rounding, v_permlane16/32_swap rounds so that a lane
//     holds 8 consecutive outputs -> 16-byte stores.  EPI 0 dense rows (+ residual), EPI 1 PF rows (+ per-image row bias + residual),
//     tokens -> PF scatter, sub-pixel scatter; EPI 2 GEGLU (weight rows packed [8 value | 8 gate] per 16-row block, value * gelu(gate));
//     EPI 3 transposed output ([B, N, S]: the attention kernel's V^T operand; MFMA operands swapped).
//   * The epilogue must not LOAD its parameters in the 8-wave variant: a vector load issued there queues behind the next tile's prefetched stages
//     (vmcnt retires in order) and the arithmetic starts a full HBM latency late.  The dense-row and GEGLU kernels stage bias / LayerNorm-fold
//     vectors / row statistics in the 4 KiB of LDS behind the ring by LDS-DMA pieces issued in the tile's first step (STG, dma_params).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "../../include/gswm.h"
#include "gswm_mm.h"
#include "gswm_mmtypes.h"
#include "gswm_ablate.inc"

// Process-wide tuning knobs of the engine (tests and A/B runs force a tiling; production leaves both on "auto").  Initialised from the environment
// (GSW_MM_BM = 128 | 256, GSW_MM_SPLIT = bit mask over epilogue kinds) and settable through the C ABI (gsw_mm_config).
static std::atomic<int> g_mm_tile_rows{getenv("GSW_MM_BM") ? atoi(getenv("GSW_MM_BM")) : 0};
static std::atomic<int> g_mm_split_mask{getenv("GSW_MM_SPLIT") ? atoi(getenv("GSW_MM_SPLIT")) : 10 /* convolutions (EPI 1) and the transposed projection (EPI 3) */};

namespace {

// measurement switches (cycle stamps, ablations): every one of them lives in gswm_ablate.inc; a production build defines none (gsw_build_flags() == 0)
__device__ unsigned long long* g_mm_trace_buf;                // (MM_TRACE builds: tools/ubench/mm_trace.hip points it at its buffer)
template <typename F>
__device__ __forceinline__ mm_f4 mm_abl_touch(F a, F b, mm_f4 c) { c[0] = fmaf((float)a[0], (float)b[0], c[0]); return c; }      // (MM_ABL_NOMFMA builds)
#define MM_BARRIER() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
// epilogue barrier: LDS image traffic only -- the DMA prefetch of the next tile stays in flight (no vmcnt wait)
#define MM_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); MM_BARRIER(); } while (0)

// LDS reads of the staged epilogue parameters as inline asm: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see while an
// LDS-DMA may be in flight (it cannot prove the destinations disjoint from the read), which is exactly the wait the staging removes.  Each helper
// ends with its own s_waitcnt lgkmcnt(0): the outputs are valid when the statement ends.
typedef float mm_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mm_lds_addr(const uint8_t* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t*)p; }
// five 8-byte cells 32 bytes apart (the bias of a lane's four columns in each of the five column blocks)
__device__ __forceinline__ void mm_lds_read5_b64(uint32_t a, uint2 (&o)[5]) {
    uint64_t r0, r1, r2, r3, r4;
    asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:32\n\tds_read_b64 %2, %5 offset:64\n\tds_read_b64 %3, %5 offset:96\n\tds_read_b64 %4, %5 offset:128\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a) : "memory");
    const uint64_t r[5] = {r0, r1, r2, r3, r4};
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = make_uint2((uint32_t)r[i], (uint32_t)(r[i] >> 32));
}
// LayerNorm fold: u and v (1 KiB apart) of a lane's four columns in each of the five column blocks, and (rstd, nrm) of its MT rows (16 rows apart)
template <uint32_t VOFF>      // VOFF: byte offset of v behind u (1 KiB; 2 KiB for the 320-column tile)
__device__ __forceinline__ void mm_lds_read_uv(uint32_t a, mm_f4 (&u)[5], mm_f4 (&v)[5]) {
    asm volatile("ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:64\n\tds_read_b128 %2, %10 offset:128\n\tds_read_b128 %3, %10 offset:192\n\tds_read_b128 %4, %10 offset:256\n\t"
                 "ds_read_b128 %5, %10 offset:%11\n\tds_read_b128 %6, %10 offset:%12\n\tds_read_b128 %7, %10 offset:%13\n\tds_read_b128 %8, %10 offset:%14\n\t"
                 "ds_read_b128 %9, %10 offset:%15\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]), "=&v"(u[4]), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                 : "v"(a), "n"(VOFF), "n"(VOFF + 64u), "n"(VOFF + 128u), "n"(VOFF + 192u), "n"(VOFF + 256u) : "memory");
}
// one column block's u and v (the wide tile reads them block by block: all five at once do not fit beside 160 accumulators)
template <uint32_t VOFF>
__device__ __forceinline__ void mm_lds_read_uv1(uint32_t a, mm_f4& u, mm_f4& v) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(u), "=&v"(v) : "v"(a), "n"(VOFF) : "memory");
}
template <int MT>
__device__ __forceinline__ void mm_lds_read_rows(uint32_t a, mm_f2 (&r)[MT]) {
    if constexpr (MT == 8)
        asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\tds_read_b64 %3, %8 offset:384\n\t"
                     "ds_read_b64 %4, %8 offset:512\n\tds_read_b64 %5, %8 offset:640\n\tds_read_b64 %6, %8 offset:768\n\tds_read_b64 %7, %8 offset:896\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]) : "v"(a) : "memory");
    else if constexpr (MT == 4)
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\tds_read_b64 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(a) : "memory");
    else
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:128\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0]), "=&v"(r[1]) : "v"(a) : "memory");
}

// lane index of the calling lane, made on the spot (exec is all ones wherever this is called): two v_mbcnt in a VOLATILE asm -- unlike threadIdx.x & 63 it needs no
// register kept alive across the main loop, and unlike the builtin it is not hoisted above it
__device__ __forceinline__ uint32_t mm_lane_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
// EPI: epilogue class the kernel is compiled for -- 0 dense rows (bias, optional residual), 1 generic (PF border / row bias / token scatter /
// sub-pixel scatter), 2 GEGLU, 3 transposed output (MFMA operands swapped); mm_gelu: gswm_mmtypes.h
// x + (the DPP-selected x of another lane; 0 where the selection has no source or the row is masked out)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mm_dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}

// SPLIT: 12 waves -- waves 0-7 only read fragments and multiply, waves 8-11 (one per SIMD) own ALL the LDS-DMA: a global_load_lds holds
// its wave for 60-185 cycles at issue, which an in-order wave that also carries MFMAs cannot hide; a producer wave can stall all it likes.
// Three waves per SIMD cap a wave at 168 registers.  (Producers that stage through registers -- 13 global_load_dwordx4 kept in flight, then
// ds_write_b128 with the swizzle on the LDS address -- were tried instead of LDS-DMA: 1022-1077 vs 1290-1358 TFLOP/s on the 3x3 shapes; the
// ds_write traffic slows the multiplying waves' fragment reads and the load latency no longer hides.  Splitting the duty -- weight pieces issued by
// the multiplying waves, activation pieces by the producers, so that all twelve waves issue -- was measured too: 1227-1268; any global_load_lds in a
// multiplying wave's stream costs more than the relief it gives the producers.)
// MT: 16-row MFMA tiles per wave along M: 4 -> the 256 x 160 tile; 2 -> a 128 x 160 tile for launches whose 256-row tiling would leave CUs without
// a tile (the 8 x 8 level at batch 64, everything deep at batch 8).
template <typename T, int EPI, bool SPLIT, int MT, bool LNF = false>
__global__ __launch_bounds__(SPLIT ? 768 : 512, SPLIT ? 1 : 2) void gsw_mm_kernel(const MMArgs p) {
    static_assert(!LNF || (!SPLIT && (EPI == 0 || EPI == 2 || EPI == 3)), "LayerNorm-folded epilogues: dense rows, GEGLU, transposed; 8-wave form");
    constexpr bool SWAP = EPI == 3;
    // EPI 5 (self-attention q | k | v from ONE pass over the tokens): column tiles below p.n_rows are dense rows (EPI 0), the others the transposed value
    // projection (EPI 3: MFMA operands swapped) -- decided per tile, the step loop exists in both forms
    constexpr bool QKV = EPI == 5;
    // EPI 4 (split-K, always the 12-wave variant): the workgroup multiplies a RANGE of its tile's K stages and dumps the fp32 accumulators into a
    // workspace slab; gsw_mm_reduce_kernel sums the slabs of a tile in a fixed order and runs the epilogue of the launch's mode
    constexpr bool PART = EPI == 4;
    static_assert(!PART || SPLIT, "split-K partial launches use the producer-wave variant");
    // WIDE (MT == 8): the 256 x 320 tile -- 8 multiplying waves as 2 (M) x 4 (N), a wave owns 128 x 80 outputs = 8 x 5 accumulators (160 registers).  Per MFMA it
    // moves 31 % fewer LDS-DMA bytes and 28 % fewer fragment bytes than the 256 x 160 tile: under the board's power limit the engine's
    // time follows its energy, and operand movement is ~40 % of it (profiles/r04h_power_cap_probe.txt section 5).  A stage is 72 KiB, so the ring has TWO slots
    // and the step is scheduled differently (step8 below): fragments are streamed (W sets double-buffered, A fragments through a ring of four) instead of
    // held twice, ONE barrier per stage sits inside the odd phase, and the DMA of stage s + 2 is issued behind it.
    constexpr bool WIDE = MT == 8;
    constexpr int WM = WIDE ? 2 : 4;                 // waves along M; NG groups of WM waves along N
    constexpr int NG = 8 / WM;
    constexpr int BM = WM * 16 * MT, BN = NG * 80;
    constexpr int NPR = MT / 2;                      // pairs of row tiles per wave (the epilogue's unit)
    static_assert(MT == 8 || MT == 4 || MT == 2, "wave tile of 128, 64 or 32 rows");
    static_assert(!WIDE || (!SPLIT && EPI != 3 && EPI != 4 && EPI != 5), "the wide tile: 8-wave form, dense rows / PF rows / GEGLU; no split-K, no transposed output, no fused q | k | v");
    constexpr int NPROD = SPLIT ? 4 : 8;             // waves that issue DMA
    constexpr int NPA = BM / 8 / NPROD;              // A pieces (8 rows x 128 B = 1 KiB) per producing wave per stage: 4 (8 with SPLIT)
    constexpr int NPW = BN / 8 / NPROD;              // full rounds of W pieces per producing wave: 2 (5 with SPLIT)
    constexpr int NEXTRA = BN / 8 - NPROD * NPW;     // waves that issue one more W piece: 4 (0 with SPLIT)
    constexpr int NDMA = NPA + NPW;                  // pieces per stage of a wave without the extra one: 6 (13 with SPLIT)
    constexpr uint32_t STAGE = (uint32_t)(BM + BN) * 128u;     // one stage = 64 k-values of every tile row: 52 KiB
    constexpr uint32_t RING = (WIDE ? 2u : 3u) * STAGE;
    constexpr int HC = 80;                           // columns owned by a group
    // STG (8-wave dense-row / GEGLU epilogues): the epilogue's per-column / per-row parameters (bias; LayerNorm fold: u, v, the rows' (rstd, nrm)) are
    // staged in the 4 KiB of LDS behind the ring by LDS-DMA pieces wave 7 issues in the tile's first step.  Fetched by the epilogue itself they are
    // vector-memory loads BEHIND the next tile's prefetched stages, i.e. the epilogue's arithmetic starts a full HBM latency late
    // (profiles/r03w_epilogue_param_wait.txt: the GEGLU launches are 6-13 % faster without a bias vector).
    // (not the LayerNorm-folded dense-row epilogue: it sits at the 256-register cap)
    constexpr bool STG = !SPLIT && !SWAP && !QKV && (EPI == 2 || (EPI == 0 && (!LNF || WIDE)));
    constexpr uint32_t PARAM = RING;                 // [0, 1 KiB): bias (fp16) or u (fp32) of the tile's 160 columns; [1, 2): v; [2, 4): (rstd, nrm) of its rows
                                                     // (WIDE: 320 columns -- u [0, 2), v [2, 4), rows [4, 6) KiB)
    constexpr uint32_t PAR_V = WIDE ? 2048u : 1024u, PAR_ROWS = WIDE ? 4096u : 2048u;
    constexpr int NPCOL = WIDE ? 2 : 1;              // LDS-DMA pieces per fp32 column vector (256 floats each)
    constexpr int NPAR = LNF ? 2 * NPCOL + (BM + 127) / 128 : 1;
    typedef typename MM<T>::frag frag;
    typedef unsigned int mm_u4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t grp = WIDE ? (wave >> 1) & 3u : (wave >> 2) & 1u, wm = WIDE ? wave & 1u : wave & 3u;
    const uint32_t pid = SPLIT ? (wave & 3u) : wave;          // index among the producing waves (SPLIT: waves 8..11)
    const bool extra = pid < (uint32_t)NEXTRA;
    // (three steps: the pieces are issued in step 0 and published by the barrier of step 2)
    // with fewer than three steps per tile the pieces are issued inside the epilogue, between two barriers
    // (WIDE: issued behind the barrier of step 0, covered by the wait in front of the barrier of step 1)
    const bool stg = STG && p.P >= (WIDE ? 2 : 3) && (LNF || p.bias != nullptr);
    const bool stg7 = stg && wave == 7u;

    // LDS rows are 128 B (64 k-values); 16-byte chunks are XOR-swizzled with (row >> 1) & 7.
    // fragment read of k-half h: row = lane & 15 of a 16-row block, logical chunk = 4h + (lane >> 4); (row >> 1) & 7 = (lane >> 1) & 7
    const uint32_t lane_rd0 = (lane & 15u) * 128u + ((((lane >> 4)) ^ ((lane >> 1) & 7u)) << 4);
    const uint32_t a_rd0 = wm * (16u * MT) * 128u + lane_rd0, a_rd1 = a_rd0 ^ 64u;            // k-half 1 = chunk + 4 = byte offset ^ 64
    const uint32_t w_rd0 = (uint32_t)BM * 128u + (grp * (uint32_t)HC) * 128u + lane_rd0, w_rd1 = w_rd0 ^ 64u;
    // DMA source: lane -> row lane >> 3 of an 8-row piece, physical chunk lane & 7 holds logical chunk (lane & 7) ^ ((row >> 1) & 7);
    // a wave's pieces are wave, wave + 8, ...: (row >> 1) & 7 = (4 (wave & 1) + (lane >> 4)) & 7
    const uint32_t prow = lane >> 3;
    const uint32_t chunk8 = ((lane & 7u) ^ ((4u * (pid & 1u) + (lane >> 4)) & 7u)) * 8u;

    // tile schedule: hardware block b runs on XCD b % 8; consecutive logical tiles (N tiles fastest) go to ONE XCD, so the
    // workgroups sharing an activation tile find it in that XCD's L2
    const uint32_t G = gridDim.x;
    const uint32_t slotx = (blockIdx.x & 7u) * (G >> 3) + (blockIdx.x >> 3);
    const uint32_t ntiles = (uint32_t)p.ntiles;
    const uint32_t nvirt = PART ? ntiles * (uint32_t)p.splits : ntiles;      // split-K: (tile, K range) pairs, at most one per workgroup
    const uint32_t nt_mine = nvirt > slotx ? (nvirt - slotx + G - 1u) / G : 0u;
    if (nt_mine == 0u) return;
    // split-K: virtual tile v = split * ntiles + tile; stages [k_lo, k_hi) of the tile's P
    const uint32_t part_split = PART ? slotx / ntiles : 0u;
    const int32_t k_lo = PART ? (int32_t)(((int64_t)part_split * p.P) / p.splits) : 0;
    const int32_t P_mine = PART ? (int32_t)(((int64_t)(part_split + 1u) * p.P) / p.splits) - k_lo : p.P;
    const T* Wbase = reinterpret_cast<const T*>(p.w);
    // logical tile -> (tile_m, tile_n): N is walked in PANELS of p.panel (8) tiles (M fastest across panels' rows), so the 32 tiles an XCD works on
    // at a time are a 4 x 8 block: 4 activation tiles + 8 weight tiles in its L2 instead of 1 + 32 for a wide projection.  Every panel streams the
    // activation matrix once more, so when ALL weight tiles of the launch fit an XCD's L2 (short K: the 64 x 64 level's GEGLU projection, 16 tiles of
    // 100 KB) the panel is the whole N and the activations are read once
    auto decode_tile = [&](uint32_t L, int32_t& tm, int32_t& tn) {
        const uint32_t tiles_m = ntiles / (uint32_t)p.tiles_n, pw = (uint32_t)p.panel;
        const uint32_t full = tiles_m * pw;
        const uint32_t pn = L / full, rem = L - pn * full;
        const uint32_t width = min(pw, (uint32_t)p.tiles_n - pn * pw);
        tm = (int32_t)(rem / width);
        tn = (int32_t)(pn * pw + rem - (rem / width) * width);
    };

    // ---------------------------------------------------------------- producer: per-lane source pointers advanced by uniform steps.
    // A RUN is a sequence of stages with constant steps: a dense segment is one run over its K blocks; a convolution segment visits
    // channel block kc (outer) x tap row kh x tap column kw (inner, the run: next pixel, next tap's weights).
    uint32_t pr_it = 0, pr_slot = 0;
    int32_t pr_seg = 0, pr_kc = 0, pr_kh = 0, pr_run = 0, a_step = 0, w_step = 0;
    int32_t pr_arow[NPA], pr_wrow[NPW + 1];
    const T* pa[NPA];
    const T* pw[NPW + 1];
    // MM_FLAG_COMPACT (convolutions): the M dimension enumerates INTERIOR pixels (b, y, x) only; the padded-flat row is computed per tile.
    // The border rows of the output are not this kernel's business then (gsw_pf_zero_border writes them).
    const bool compact = (p.flags & MM_FLAG_COMPACT) != 0;
    const int32_t HpWp = p.Hp * p.Wp;
    // (WIDE: a divisor goes through an empty asm where it is used -- hipcc otherwise hoists the reciprocal it divides with above the tile loop, one VGPR per
    // divisor kept alive, i.e. spilled, across the main loop)
    auto opaque = [&](int32_t v) -> int32_t { if constexpr (WIDE) asm volatile("" : "+s"(v)); return v; };
    // WIDE: the two divisions by wave-uniform divisors as multiply-high by floor((2^32 - 1) / d) plus ONE correction step (the estimate is the quotient or one
    // less for any 32-bit dividend): six VALU instructions and two temporaries instead of hipcc's ~20-instruction float-reciprocal sequence, whose temporaries
    // do not fit beside 160 accumulators and the fragment sets where the producer's slow path runs (the tail of an odd phase)
    // (the two multipliers are recomputed where they are used -- a scalar division in a slow path -- rather than kept: the scalar registers are all taken too)
    auto udiv = [&](uint32_t m, uint32_t d, uint32_t mg) -> uint32_t {
        uint32_t q = __umulhi(m, mg);
        if (m - q * d >= d) ++q;
        return q;
    };
    auto pf_row = [&](int32_t m, int32_t& b, int32_t& yy, int32_t& xx) -> int32_t {      // interior index -> (image, padded y, padded x), row
        if constexpr (WIDE) {
            const int32_t Wi = opaque(p.Wp - 2), HW = opaque((p.Hp - 2) * (p.Wp - 2));
            const uint32_t mg_hw = 0xFFFFFFFFu / (uint32_t)HW, mg_wi = 0xFFFFFFFFu / (uint32_t)Wi;
            b = (int32_t)udiv((uint32_t)m, (uint32_t)HW, mg_hw);
            const int32_t r = m - b * HW;
            yy = (int32_t)udiv((uint32_t)r, (uint32_t)Wi, mg_wi);
            xx = r - yy * Wi + 1;
            yy += 1;
            return b * HpWp + yy * p.Wp + xx;
        }
        const int32_t Wi = p.Wp - 2, HW = (p.Hp - 2) * Wi;
        b = m / HW;
        const int32_t r = m - b * HW;
        yy = r / Wi;
        xx = r - yy * Wi + 1;
        yy += 1;
        return b * HpWp + yy * p.Wp + xx;
    };
    auto setup_tile = [&](uint32_t it) {
        int32_t tm, tn;
        decode_tile(PART ? slotx % ntiles : it * G + slotx, tm, tn);
        const int32_t m0 = tm * BM, n0 = tn * BN;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int32_t m = m0 + 8 * (int32_t)(pid + (uint32_t)NPROD * i) + (int32_t)prow;
            const int32_t mc = m < p.M ? m : p.M - 1;
            int32_t b_, y_, x_;
            pr_arow[i] = compact ? pf_row(mc, b_, y_, x_) : mc;
            // stride 2: the window of output pixel (y, x) starts at padded input coordinates (2 (y-1), 2 (x-1)); taps add kh * in_Wp + kw
            if (compact && p.stride == 2) pr_arow[i] = b_ * (p.in_Hp * p.in_Wp) + 2 * (y_ - 1) * p.in_Wp + 2 * (x_ - 1);
        }
#pragma unroll
        for (int i = 0; i <= NPW; ++i) {
            const int32_t n = n0 + 8 * (int32_t)(pid + (uint32_t)NPROD * i) + (int32_t)prow;
            pr_wrow[i] = n < p.N ? n : p.N - 1;              // (only the unused extra piece of waves >= NEXTRA can exceed the tile)
        }
    };
    auto begin_run = [&]() {
        const MMSeg& s0 = p.seg[0]; const MMSeg& s1 = p.seg[1]; const MMSeg& s2 = p.seg[2];
        const int32_t s = pr_seg;
        const T* x = reinterpret_cast<const T*>(s == 0 ? s0.x : s == 1 ? s1.x : s2.x);
        const int32_t ld = s == 0 ? s0.ld : s == 1 ? s1.ld : s2.ld;
        const int32_t kb = s == 0 ? s0.kblocks : s == 1 ? s1.kblocks : s2.kblocks;
        const int32_t nt = s == 0 ? s0.ntaps : s == 1 ? s1.ntaps : s2.ntaps;
        const int32_t tw = s == 0 ? s0.tw : s == 1 ? s1.tw : s2.tw;
        const int32_t trow = s == 0 ? s0.tap_row : s == 1 ? s1.tap_row : s2.tap_row;
        const int32_t tbase = s == 0 ? s0.tap_base : s == 1 ? s1.tap_base : s2.tap_base;
        const int32_t wk0 = s == 0 ? s0.wk0 : s == 1 ? s1.wk0 : s2.wk0;
        const int32_t tapoff = tbase + pr_kh * trow;
        const int32_t wk = wk0 + pr_kh * tw * (kb * 64) + pr_kc * 64 + (int32_t)chunk8;
        if (nt == 1) { pr_run = kb; a_step = 64; w_step = 64; }
        else { pr_run = tw; a_step = ld; w_step = kb * 64; }
#pragma unroll
        for (int i = 0; i < NPA; ++i) pa[i] = x + ((int64_t)(pr_arow[i] + tapoff) * ld + pr_kc * 64 + (int32_t)chunk8);
#pragma unroll
        for (int i = 0; i <= NPW; ++i) pw[i] = Wbase + (pr_wrow[i] * p.ldw + wk);
    };
    auto end_run = [&]() {                                   // slow path: next tap row / channel block / segment / tile
        const MMSeg& s0 = p.seg[0]; const MMSeg& s1 = p.seg[1]; const MMSeg& s2 = p.seg[2];
        const int32_t s = pr_seg;
        const int32_t kb = s == 0 ? s0.kblocks : s == 1 ? s1.kblocks : s2.kblocks;
        const int32_t nt = s == 0 ? s0.ntaps : s == 1 ? s1.ntaps : s2.ntaps;
        const int32_t tw = s == 0 ? s0.tw : s == 1 ? s1.tw : s2.tw;
        bool seg_done = nt == 1;
        if (!seg_done && ++pr_kh * tw == nt) { pr_kh = 0; seg_done = ++pr_kc == kb; }
        if (seg_done) {
            pr_kc = 0; pr_kh = 0;
            if (++pr_seg == p.nseg) {
                pr_seg = 0;
                if (++pr_it == nt_mine) {
                    // no stage left: the (unconditional) DMA of the remaining steps re-reads one valid, cached location into ring slots
                    // that are never read again, so the loop body needs no "is there a stage" branch and the wait counts stay exact
                    pr_run = 0x7FFFFFFF; a_step = 0; w_step = 0;
#pragma unroll
                    for (int i = 0; i < NPA; ++i) pa[i] = Wbase + chunk8;
#pragma unroll
                    for (int i = 0; i <= NPW; ++i) pw[i] = Wbase + chunk8;
                    return;
                }
                setup_tile(pr_it);
            }
        }
        begin_run();
    };
    // A stage's DMA is issued in two halves, half a step apart, each interleaved sparsely (one piece per three MFMAs) with a phase's MFMAs:
    // a glds occupies the CU's address unit for ~17 cycles and the issuing wave until it is accepted, so with one MFMA per piece the
    // matrix pipe starves (measured: the whole 52-piece burst exposed, tools/ubench/mm_trace).
    //   H1(stage j) = activation pieces 0, 1 + weight piece 0          -- odd phase of step j-3
    //   H2(stage j) = activation pieces 2, 3 + weight piece 1 (+ 2)    -- even phase of step j-2
    auto dma_extra = [&]() {                                 // the W piece only waves < NEXTRA carry (wave-uniform branch, kept out of the main block)
        if (extra) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pw[NPW],
                                             (__attribute__((address_space(3))) void*)(lds + pr_slot + (uint32_t)BM * 128u + (pid + (uint32_t)NPROD * NPW) * 1024u), 16, 0, 0);
            pw[NPW] += w_step;
        }
    };
    auto dma_piece_a = [&](int i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[i],
                                         (__attribute__((address_space(3))) void*)(lds + pr_slot + (pid + (uint32_t)NPROD * i) * 1024u), 16, 0, 0);
        pa[i] += a_step;
    };
    auto dma_piece_w = [&](int i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pw[i],
                                         (__attribute__((address_space(3))) void*)(lds + pr_slot + (uint32_t)BM * 128u + (pid + (uint32_t)NPROD * i) * 1024u), 16, 0, 0);
        pw[i] += w_step;
    };
    // ---------------------------------------------------------------- WIDE producer: the same walk with BUFFER addressing.  160 accumulators leave no room for
    // nine 64-bit lane pointers + nine row indices: a piece's address is (buffer base in SGPRs) + (32-bit lane offset) + (scalar offset), and what a stage
    // advances is the scalar offset only.  (global_load_lds on base + 32-bit offsets selects the saddr form and needs no descriptor, but hipcc's waitcnt pass
    // then turns every counted lgkmcnt wait of the fragment reads into lgkmcnt(0): measured in the ISA, not adopted.)  Activations: one lane offset per piece (row of this lane x row stride + its swizzled chunk), recomputed when the
    // segment changes (the row stride does); the segment's tap base is folded into the buffer base so every offset stays non-negative.  Weights: ONE lane offset
    // for the whole kernel ((8 pid + lane row) rows + chunk); tile column, piece (64 rows apart), tap and channel block live in the scalar offset.  Needs
    // N % 320 == 0 (no weight-row clamp) and operands below 4 GiB (host-checked).
    // EPI 0 / 2 (dense operands only, M % 256 == 0 host-checked): the four activation pieces are 64 rows apart too -- one lane offset, scalar piece offsets
    constexpr bool AFF = EPI == 0 || EPI == 2;
    uint32_t vo_a[4] = {0u, 0u, 0u, 0u}, vo_w = 0u, so_a = 0u, so_w = 0u, st_a = 0u, st_w = 0u, w_piece = 0u, a_piece = 0u;
    int32_t seg_rows = -1;                                    // segment vo_a was computed for
    bool pr_end = false;                                      // a stage completed its run: end_run8() pending
    __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, -1, 0x00020000);
    // byte strides between the 64-row pieces of a stage, pinned in scalar registers: read from the kernel arguments where they are used, hipcc re-loads them
    // with s_load INSIDE the main loop, and a scalar load in flight turns every counted lgkmcnt wait of the fragment reads into lgkmcnt(0)
    uint32_t str_a8 = 128u * (uint32_t)p.seg[0].ld, str_w8 = 128u * (uint32_t)p.ldw;
    if constexpr (WIDE) asm volatile("" : "+s"(str_a8), "+s"(str_w8));
    int32_t tile_m8 = 0, tile_n8 = 0;
    auto setup_tile8 = [&](uint32_t it) {
        decode_tile(it * G + slotx, tile_m8, tile_n8);
        seg_rows = -1;
    };
    auto begin_run8 = [&]() {
        const MMSeg& s0 = p.seg[0]; const MMSeg& s1 = p.seg[1]; const MMSeg& s2 = p.seg[2];
        const int32_t sg = pr_seg;
        const T* x = reinterpret_cast<const T*>(sg == 0 ? s0.x : sg == 1 ? s1.x : s2.x);
        const int32_t ld = sg == 0 ? s0.ld : sg == 1 ? s1.ld : s2.ld;
        const int32_t kb = sg == 0 ? s0.kblocks : sg == 1 ? s1.kblocks : s2.kblocks;
        const int32_t nt = sg == 0 ? s0.ntaps : sg == 1 ? s1.ntaps : s2.ntaps;
        const int32_t tw = sg == 0 ? s0.tw : sg == 1 ? s1.tw : s2.tw;
        const int32_t trow = sg == 0 ? s0.tap_row : sg == 1 ? s1.tap_row : s2.tap_row;
        const int32_t tbase = sg == 0 ? s0.tap_base : sg == 1 ? s1.tap_base : s2.tap_base;
        const int32_t wk0 = sg == 0 ? s0.wk0 : sg == 1 ? s1.wk0 : s2.wk0;
        if (seg_rows != sg) {                                 // new tile or new segment: lane offsets of the four activation pieces
            seg_rows = sg;
            // (lane row and swizzled chunk are recomputed HERE, behind an empty asm: kept from the kernel's start they are long-lived values the
            // allocator spills around the main loop)
            const uint32_t l8 = mm_lane_now();

            const uint32_t prow = l8 >> 3;
            const uint32_t chunk8 = ((l8 & 7u) ^ ((4u * (pid & 1u) + (l8 >> 4)) & 7u)) * 8u;
            rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (int64_t)tbase * ld), 0, -1, 0x00020000);
            const int32_t m0 = tile_m8 * BM;
            if constexpr (AFF) {
                vo_a[0] = ((uint32_t)(m0 + 8 * (int32_t)pid + (int32_t)prow) * (uint32_t)ld + chunk8) * 2u;
            } else {
#pragma unroll
                for (int i = 0; i < NPA; ++i) {
                    const int32_t m = m0 + 8 * (int32_t)(pid + (uint32_t)NPROD * i) + (int32_t)prow;
                    const int32_t mc = m < p.M ? m : p.M - 1;
                    int32_t b_, y_, x_, row = mc;
                    if (compact) {
                        row = pf_row(mc, b_, y_, x_);
                        if (p.stride == 2) row = b_ * (p.in_Hp * p.in_Wp) + 2 * (y_ - 1) * p.in_Wp + 2 * (x_ - 1);
                    }
                    vo_a[i] = ((uint32_t)row * (uint32_t)ld + chunk8) * 2u;
                }
            }
        }
        so_a = (uint32_t)(pr_kh * trow * ld + pr_kc * 64) * 2u;
        so_w = (uint32_t)(tile_n8 * BN * p.ldw + wk0 + pr_kh * tw * (kb * 64) + pr_kc * 64) * 2u;
        if (nt == 1) { pr_run = kb; st_a = 128u; st_w = 128u; }
        else { pr_run = tw; st_a = (uint32_t)ld * 2u; st_w = (uint32_t)(kb * 64) * 2u; }
    };
    auto end_run8 = [&]() {                                  // slow path: next tap row / channel block / segment / tile
        const MMSeg& s0 = p.seg[0]; const MMSeg& s1 = p.seg[1]; const MMSeg& s2 = p.seg[2];
        const int32_t sg = pr_seg;
        const int32_t kb = sg == 0 ? s0.kblocks : sg == 1 ? s1.kblocks : s2.kblocks;
        const int32_t nt = sg == 0 ? s0.ntaps : sg == 1 ? s1.ntaps : s2.ntaps;
        const int32_t tw = sg == 0 ? s0.tw : sg == 1 ? s1.tw : s2.tw;
        bool seg_done = nt == 1;
        if (!seg_done && ++pr_kh * tw == nt) { pr_kh = 0; seg_done = ++pr_kc == kb; }
        if (seg_done) {
            pr_kc = 0; pr_kh = 0;
            if (++pr_seg == p.nseg) {
                pr_seg = 0;
                if (++pr_it == nt_mine) {
                    // no stage left: the (unconditional) DMA of the remaining steps re-reads ONE valid, cached location (the first K block of the lanes' rows,
                    // the first 320 weight rows) into ring slots nobody reads.  (Not "where the offsets stand": they were already advanced past the last stage --
                    // for the last weight rows that is past the end of the matrix.)
                    pr_run = 0x7FFFFFFF; st_a = 0u; st_w = 0u;
                    so_a = 0u; so_w = 0u;
                    return;
                }
                setup_tile8(pr_it);
            }
        }
        begin_run8();
    };
    // piece j = 0 .. 8 of the stage being issued (four activation pieces, then five weight pieces 64 rows apart); the last one completes the stage
    auto dma_piece8 = [&](int j) {
        if (j < NPA) {
            if constexpr (AFF) {
                if (j == 0) a_piece = so_a;
                MM_ABL_DMA_A(__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds + pr_slot + (pid + (uint32_t)NPROD * j) * 1024u), 16, (int)vo_a[0], (int)a_piece, 0, 0));
                a_piece += str_a8;
            } else {
                MM_ABL_DMA_A(__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(lds + pr_slot + (pid + (uint32_t)NPROD * j) * 1024u), 16, (int)vo_a[j], (int)so_a, 0, 0));
            }
        } else {
            if (j == NPA) w_piece = so_w;
            MM_ABL_DMA_W(__builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(lds + pr_slot + (uint32_t)BM * 128u + (pid + (uint32_t)NPROD * (j - NPA)) * 1024u), 16,
                                                                  (int)vo_w, (int)w_piece, 0, 0));
            w_piece += str_w8;
        }
        if (j == NDMA - 1) {
            so_a += st_a; so_w += st_w;
            pr_slot = pr_slot + STAGE == RING ? 0u : pr_slot + STAGE;
            pr_end = --pr_run == 0;          // the run is exhausted: end_run8() is due before the next stage is issued (the caller decides where -- there are three
                                             // issue sites and the slow path is inlined at each place that calls it: kept to two)
        }
    };
    auto dma_h1 = [&]() {
#pragma unroll
        for (int i = 0; i < NPA / 2; ++i) dma_piece_a(i);
        dma_piece_w(0);
    };
    auto dma_h2 = [&]() {                                    // completes the stage: the ring slot advances
#pragma unroll
        for (int i = NPA / 2; i < NPA; ++i) dma_piece_a(i);
        dma_piece_w(1);
        pr_slot = pr_slot + STAGE == RING ? 0u : pr_slot + STAGE;
    };
    if constexpr (SPLIT) {
        if (wave >= 8u) {
            // ------------------------------------------------------------ producer wave: every stage, all 13 of its pieces back to back.
            // Barrier k (k = 0 .. stages) is the consumers' "stage k is in LDS and the slot of stage k-1 is free": stage s+2 is issued after
            // barrier s-1... i.e. right after the barrier that retired the slot's previous tenant, and stage s+1 has landed before barrier s.
            int32_t pr_left = 0x7FFFFFFF;                                    // (PART) stages of this workgroup's K range still to issue
            auto dma_stage = [&]() {
#pragma unroll
                for (int i = 0; i < NPA; ++i) MM_ABL_DMA_A(dma_piece_a(i));
#pragma unroll
                for (int i = 0; i < NPW; ++i) MM_ABL_DMA_W(dma_piece_w(i));
                pr_slot = pr_slot + STAGE == RING ? 0u : pr_slot + STAGE;
                if constexpr (PART) {
                    if (--pr_left == 0) {                        // the K range of this workgroup is issued: filler reads from here on (see end_run)
                        pr_run = 0x7FFFFFFF; a_step = 0; w_step = 0;
#pragma unroll
                        for (int i = 0; i < NPA; ++i) pa[i] = Wbase + chunk8;
#pragma unroll
                        for (int i = 0; i <= NPW; ++i) pw[i] = Wbase + chunk8;
                        return;
                    }
                }
                if (--pr_run == 0) end_run();
            };
            constexpr int NWAIT = MM_ABL_NPA(NPA) + MM_ABL_NPW(NPW);             // pieces of a stage this wave has in flight (= NDMA in a production build)
            setup_tile(0);
            if constexpr (PART) {
                // enter the tile's stage sequence at stage k_lo: segment, channel block, tap row, and the position inside the run
                const MMSeg& s0 = p.seg[0]; const MMSeg& s1 = p.seg[1];
                int32_t st = k_lo;
                const int32_t len0 = s0.kblocks * s0.ntaps, len1 = s1.kblocks * s1.ntaps;
                if (p.nseg > 1 && st >= len0) { st -= len0; pr_seg = 1; if (p.nseg > 2 && st >= len1) { st -= len1; pr_seg = 2; } }
                const int32_t nt = pr_seg == 0 ? s0.ntaps : 1, tw = pr_seg == 0 ? s0.tw : 1;      // segments 1 and 2 are 1x1 (one tap)
                int32_t skip = st;
                if (nt != 1) { pr_kc = st / nt; const int32_t r = st - pr_kc * nt; pr_kh = r / tw; skip = r - pr_kh * tw; }
                begin_run();
#pragma unroll
                for (int i = 0; i < NPA; ++i) pa[i] += (int64_t)skip * a_step;
#pragma unroll
                for (int i = 0; i <= NPW; ++i) pw[i] += (int64_t)skip * w_step;
                pr_run -= skip;
                pr_left = P_mine;
            } else {
                begin_run();
            }
            dma_stage();
            dma_stage();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");         // stage 0 has landed
            MM_BARRIER();
            const uint32_t total = nt_mine * (uint32_t)P_mine;
            MM_TRACE_DECL();
            for (uint32_t sidx = 0; sidx < total; ++sidx) {
                dma_stage();                                                     // stage sidx + 2 (a cached dummy location once the work is issued)
                MM_STAMP(0);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NWAIT) : "memory");     // stage sidx + 1 has landed
                MM_STAMP(1);
                MM_BARRIER();
                MM_STAMP(2);
            }
            MM_TRACE_DUMP();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // nothing may land in LDS after the workgroup is gone
            return;
        }
    } else if constexpr (WIDE) {
        // prologue: stages 0 and 1 (both slots are free); the first even phase of a tile issues nothing (below)
        {
            const uint32_t l8 = mm_lane_now();

            vo_w = (((8u * pid + (l8 >> 3)) * (uint32_t)p.ldw) + ((l8 & 7u) ^ ((4u * (pid & 1u) + (l8 >> 4)) & 7u)) * 8u) * 2u;
        }
        setup_tile8(0);
        begin_run8();
#pragma unroll
        for (int j = 0; j < NDMA; ++j) dma_piece8(j);
        if (pr_end) { end_run8(); pr_end = false; }
#pragma unroll
        for (int j = 0; j < NDMA; ++j) dma_piece8(j);          // (its end_run8, if due, runs behind the first barrier)
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0) as a builtin: hipcc's waitcnt pass must know that no LDS-DMA is in flight when the tile loop is entered
    } else {
        static_assert(SPLIT || ((NPA == 4 || NPA == 2) && NPW == 2), "the half-stage split assumes 4 (2) + 2 (+1) pieces per wave");
        // prologue: stages 0 and 1 and the first half of stage 2; wait for stage 0 (counted: the newer pieces stay in flight; exact for waves
        // without the extra piece, conservative for the others; any other VMEM operation in flight only makes a wait longer)
        setup_tile(0);
        begin_run();
        for (int i = 0; i < 2; ++i) { dma_h1(); dma_extra(); dma_h2(); if (--pr_run == 0) end_run(); }
        dma_h1();
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA + NPA / 2 + 1) : "memory");
    }
    MM_BARRIER();

    // ---------------------------------------------------------------- consumer: a STEP = one stage = two phases of 32 k-values; every wave:
    //   even phase: DMA(stage s+2) | fragment reads (s, k-half 1) into the other register set | 20 MFMAs of (s, k-half 0) | counted wait | barrier
    //   odd phase :                  fragment reads (s+1, k-half 0)                            | 20 MFMAs of (s, k-half 1)                | barrier
    const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
    const uint16_t* rowbias = reinterpret_cast<const uint16_t*>(p.rowbias);
    const uint16_t* resid = reinterpret_cast<const uint16_t*>(p.resid);
    uint16_t* Y = reinterpret_cast<uint16_t*>(p.y);
    mm_f4 acc[5][MT];
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = mm_f4{0.f, 0.f, 0.f, 0.f};
    uint32_t c_it = 0, rd_slot = 0;                           // rd_slot: ring offset of the stage being multiplied
    MM_TRACE_DECL();

    auto read_frags = [&](frag (&xf)[MT], frag (&wf)[5], uint32_t slot, uint32_t khalf) {
        const uint8_t* ap = lds + ((khalf ? a_rd1 : a_rd0) + slot);
        const uint8_t* wp = lds + ((khalf ? w_rd1 : w_rd0) + slot);
#pragma unroll
        for (int im = 0; im < MT; ++im) xf[im] = *reinterpret_cast<const frag*>(ap + im * 2048);
#pragma unroll
        for (int in = 0; in < 5; ++in) wf[in] = *reinterpret_cast<const frag*>(wp + in * 2048);
    };

    auto dma_params = [&]() {
        int32_t tm, tn;
        decode_tile(c_it * G + slotx, tm, tn);
        const int32_t m0 = tm * BM, n0 = tn * BN;
        const uint32_t lane_p = WIDE ? mm_lane_now() : (threadIdx.x & 63u);      // (WIDE: nothing derived from the lane index may be hoisted into the main loop's registers)
        const uint32_t lane = lane_p;
        // one 16-byte-per-lane LDS-DMA piece: `base` (wave-uniform) + this lane's byte offset -> the parameter area.  WIDE: the buffer form -- a global_load_lds
        // inside the step loop (even behind a branch one wave takes once per tile) makes hipcc's waitcnt pass turn every counted lgkmcnt wait of the fragment
        // reads into lgkmcnt(0) (seen in the ISA of the dense-row kernel); the buffer form does not
        auto piece = [&](const void* base, uint32_t lane_bytes, uint32_t byte_off) {
            if constexpr (WIDE) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + PARAM + byte_off), 16, (int)lane_bytes, 0, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const uint8_t*>(base) + lane_bytes),
                                                 (__attribute__((address_space(3))) void*)(lds + PARAM + byte_off), 16, 0, 0);
            }
        };
        // lanes past the end of a vector / of the rows re-read its last 16 bytes: their LDS positions belong to columns / rows that are never stored
        if constexpr (LNF) {
#pragma unroll
            for (int k = 0; k < NPCOL; ++k) {
                const uint32_t nc = (uint32_t)min(n0 + 256 * k + 4 * (int32_t)lane, p.N - 4) * 4u;
                piece(p.ln_u, nc, (uint32_t)k * 1024u);
                piece(p.ln_v, nc, PAR_V + (uint32_t)k * 1024u);
            }
#pragma unroll
            for (int k = 0; k < (BM + 127) / 128; ++k) piece(p.ln_stat, (uint32_t)min(m0 + 128 * k + 2 * (int32_t)lane, p.M - 2) * 8u, PAR_ROWS + (uint32_t)k * 1024u);
        } else {
            piece(p.bias, (uint32_t)min(n0 + 8 * (int32_t)lane, p.N - 8) * 2u, 0u);        // 512 columns per piece: one covers either tile
        }
    };

    // ---------------------------------------------------------------- epilogue: straight from the accumulators, no LDS image, no barrier.
    // MFMA layout (non-SWAP): lane (q = lane >> 4, i = lane & 15) holds columns 16 in + 4 q + j (j = 0..3) of row 16 im + i.  After
    // bias + rounding + packing (4 fp16 = 2 registers per (in, im)), ONE v_permlane16_swap per register between row tiles im = 2p and
    // 2p + 1 leaves every lane with 8 CONSECUTIVE columns (16 bytes) of one row:
    //     row tile 2p + (q & 1), columns 16 in + 8 (q >> 1) .. + 7
    // so a wave stores its 64 x 80 block with 10 global_store_dwordx4 per lane (32 contiguous bytes per row per instruction; a row's five
    // segments come from the same wave back to back and merge in L2).  Residual / row-bias operands are loaded with the same addressing.
    // SWAP (transposed output): the same with rows and columns exchanged -- a lane ends with 8 consecutive TOKENS of one output row.
    // GEGLU: weight rows are packed [8 value | 8 gate] per 16-row block (pf.pack_geglu_weight), so value (q < 2) and gate (q >= 2) of one
    // output sit in lanes l and l + 32 of the SAME accumulator register: v_permlane32_swap pairs them, all 64 lanes compute value *
    // gelu(gate), two more swap rounds collect 8 consecutive outputs per lane.
    auto epilogue = [&]() {
        int32_t tile_m, tile_n;
        decode_tile(PART ? slotx % ntiles : c_it * G + slotx, tile_m, tile_n);
        const int32_t m0 = tile_m * BM, n0 = tile_n * BN;
        // (WIDE: the lane index goes through an empty asm so that nothing derived from it -- row / column offsets of ten stores -- is hoisted above the tile
        // loop: the main loop has no register to spare for it)
        const uint32_t lane_e = WIDE ? mm_lane_now() : (threadIdx.x & 63u);

        const uint32_t lane = lane_e;
        const uint32_t q = lane >> 4, li = lane & 15u;
        // zero-extension of a lane-dependent 32-bit offset to 64 bits with a zero made HERE (volatile asm: not hoisted): hipcc otherwise keeps ONE zero register for all
        // such pairs, hoists it above the tile loop, and -- on the wide tile, where the main loop needs every register -- spills it around the loop
        uint32_t zero_hi = 0u;
        if constexpr (WIDE) asm volatile("v_mov_b32 %0, 0" : "=v"(zero_hi));
        auto zx = [&](uint32_t v) -> int64_t { return WIDE ? (int64_t)(((uint64_t)zero_hi << 32) | (uint64_t)v) : (int64_t)v; };
        MM_STAMP(13);
        // output row of tile row m for the PF-row epilogues (EPI 1): padded-flat row of the interior pixel / of the up-sampled pixel / of the token; image, border flag
        auto out_row = [&](int32_t m, int32_t& img, bool& brd, bool& lv) -> int64_t {
            lv = m < p.M; brd = false; img = 0;
            int64_t orow_ = m;
            const int32_t mc = lv ? m : 0;
            if (p.mode == MM_MODE_PF || p.mode == MM_MODE_UP2X) {
                int32_t b, yy, xx;
                if (compact) {
                    orow_ = pf_row(mc, b, yy, xx);
                } else {
                    const int32_t d1 = opaque(HpWp), d2 = opaque(p.Wp);
                    b = mc / d1;
                    const int32_t r = mc - b * HpWp;
                    yy = r / d2; xx = r - yy * p.Wp;
                    brd = (yy == 0) | (yy == p.Hp - 1) | (xx == 0) | (xx == p.Wp - 1);
                }
                img = b;
                if (p.mode == MM_MODE_UP2X) {             // low-resolution pixel (yy-1, xx-1) -> pixel (2(yy-1)+dy, 2(xx-1)+dx) of a [B, 2H+2, 2W+2] PF tensor
                    const int32_t dy = (p.up - 1) >> 1, dx = (p.up - 1) & 1;
                    orow_ = ((int64_t)b * (2 * p.Hp - 2) + (2 * (yy - 1) + dy + 1)) * (2 * p.Wp - 2) + (2 * (xx - 1) + dx + 1);
                    lv = lv && !brd;
                }
            } else if (p.mode == MM_MODE_TOK2PF) {        // token (b, y, x) -> interior row of the PF tensor [B, H+2, W+2]
                const int32_t d1 = opaque(p.S), d2 = opaque(p.Wimg);
                const int32_t b = mc / d1, ii = mc - b * p.S;
                const int32_t yy = ii / d2, xx = ii - yy * p.Wimg;
                img = b;
                orow_ = (int64_t)b * HpWp + (int64_t)(yy + 1) * p.Wp + (xx + 1);
            }
            return orow_;
        };
        // WIDE PF rows, launches with a residual operand (dense rows: the queue of RD chunk loads further down): the epilogue walks 20 (column block, row pair) blocks per wave, each a load -> add -> store chain, and with 160
        // accumulators live it can keep ONE block's load in flight -- twenty HBM round trips in a row, ~12 us per tile with the matrix pipe idle (the fixed cost per
        // launch of the first build: 293 us against 194 for the narrow tile on the 64 x 64 convolutions).  So the wave first TOUCHES every 128-byte line of its
        // 128 x 80 residual block: four 4-byte LDS-DMA loads (64 lines each, no destination register; they land in a dummy LDS area) bring the lines into L2 / L1 in
        // ONE round trip, and the chunk loads behind them hit.
        if constexpr (WIDE && EPI == 1) {
            if (resid) {
                // (buffer form like every LDS-DMA of the wide kernel: see dma_params; the residual tensor is below 4 GiB, host-checked)
                const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(resid), 0, -1, 0x00020000);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)k * 32u + (lane >> 1));
                    int32_t img_; bool brd_, lv_ = m < p.M;
                    int64_t row_ = m;
                    if constexpr (EPI == 1) row_ = out_row(m, img_, brd_, lv_);
                    const int64_t col_ = (int64_t)n0 + grp * (uint32_t)HC + (lane & 1u) * 64u;
                    if (lv_ && col_ < p.N)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (__attribute__((address_space(3))) void*)(lds + RING + 6144u), 4, (int)(uint32_t)((row_ * p.ldr + col_) * 2), 0, 0, 0);
                }
            }
        }
        const bool vtile = QKV && n0 >= p.n_rows;                 // (wave-uniform) this tile belongs to the transposed part
        const bool transposed = SWAP || vtile;
        if constexpr (PART) {
            // slab of virtual tile v: [wave][5 * MT accumulators][lane] float4 -- one coalesced 1 KiB store per accumulator; 16-row blocks past M are
            // neither stored nor read back
            // (the slot index goes through an empty asm so that the address arithmetic stays HERE: hoisted above the loop it cost the 256-row variant, at the
            // 168-register cap of the 12-wave form, six spilled address pairs; wave-uniform base + 32-bit lane offset)
            uint32_t sx = slotx * 8u + wave;
            asm volatile("" : "+s"(sx));
            float* slab = p.ws + (size_t)sx * (size_t)(5 * MT * 64 * 4);
            const uint32_t loff = lane * 4u;
#pragma unroll
            for (int in = 0; in < 5; ++in)
#pragma unroll
                for (int im = 0; im < MT; ++im)
                    if (m0 + (int32_t)(wm * (16u * MT)) + im * 16 < p.M)
                        *reinterpret_cast<mm_f4*>(slab + (uint32_t)((in * MT + im) * 256) + loff) = acc[in][im];
            ++c_it;
            return;
        }
        // LNF: per-row (rstd, nrm) of this lane's rows and the fp32 column vectors u, v replace the bias: value = rstd acc + nrm u + v
        float ln_r[SWAP ? 4 * MT : MT], ln_n[SWAP ? 4 * MT : MT];
        mm_f4 ln_u4[5], ln_v4[5];
        if constexpr (LNF) {
            const float2* st = reinterpret_cast<const float2*>(p.ln_stat);
            if constexpr (!SWAP) {
                if constexpr (STG) {                               // staged behind the ring by wave 7 (dma_params)
                    if (!stg) {                                    // (short tiles: now)
                        MM_BARRIER();
                        if (wave == 7u) { dma_params(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                        MM_BARRIER();
                    }
                    if constexpr (!(WIDE && EPI == 0)) {           // (WIDE dense rows: lnf_rows, per row pair)
                        mm_f2 rw[MT];
                        mm_lds_read_rows<MT>(mm_lds_addr(lds + PARAM + PAR_ROWS) + (wm * (16u * MT) + li) * 8u, rw);
#pragma unroll
                        for (int im = 0; im < MT; ++im) { ln_r[im] = rw[im][0]; ln_n[im] = rw[im][1]; }
                    }
                    if constexpr (!WIDE) mm_lds_read_uv<PAR_V>(mm_lds_addr(lds + PARAM) + (grp * HC + q * 4u) * 4u, ln_u4, ln_v4);      // (WIDE: lnf_cols, per block)
                } else {
#pragma unroll
                for (int im = 0; im < MT; ++im) {
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)im * 16u + li);
                    const float2 v = st[m < p.M ? m : p.M - 1];
                    ln_r[im] = v.x; ln_n[im] = v.y;
                }
#pragma unroll
                for (int in = 0; in < 5; ++in) {
                    const int32_t nb = n0 + (int32_t)(grp * HC + (uint32_t)in * 16u + q * 4u);
                    const int32_t nc = nb < p.N ? nb : 0;
                    ln_u4[in] = *reinterpret_cast<const mm_f4*>(p.ln_u + nc);
                    ln_v4[in] = *reinterpret_cast<const mm_f4*>(p.ln_v + nc);
                }
                }
            } else {
#pragma unroll
                for (int im = 0; im < MT; ++im) {
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)im * 16u + q * 4u);       // four consecutive tokens (M % 8 == 0)
                    const int32_t mc = m + 3 < p.M ? m : 0;
                    const mm_f4 a = *reinterpret_cast<const mm_f4*>(st + mc), b = *reinterpret_cast<const mm_f4*>(st + mc + 2);
                    ln_r[4 * im] = a[0]; ln_n[4 * im] = a[1]; ln_r[4 * im + 1] = a[2]; ln_n[4 * im + 1] = a[3];
                    ln_r[4 * im + 2] = b[0]; ln_n[4 * im + 2] = b[1]; ln_r[4 * im + 3] = b[2]; ln_n[4 * im + 3] = b[3];
                }
#pragma unroll
                for (int in = 0; in < 5; ++in) {
                    const int32_t ncol = n0 + (int32_t)(grp * HC + (uint32_t)in * 16u + li);
                    const int32_t nc = ncol < p.N ? ncol : 0;
                    ln_u4[in] = mm_f4{p.ln_u[nc], 0.f, 0.f, 0.f};
                    ln_v4[in] = mm_f4{p.ln_v[nc], 0.f, 0.f, 0.f};
                }
            }
        }
        // WIDE dense rows: (rstd, nrm) of the two row tiles of row pair `pr`, read when the pair is processed
        auto lnf_rows = [&](int pr) {
            if constexpr (LNF && WIDE && STG && EPI == 0) {
                mm_f2 rw[2];
                mm_lds_read_rows<2>(mm_lds_addr(lds + PARAM + PAR_ROWS) + (wm * (16u * MT) + (uint32_t)(2 * pr) * 16u + li) * 8u, rw);
                ln_r[2 * pr] = rw[0][0]; ln_n[2 * pr] = rw[0][1]; ln_r[2 * pr + 1] = rw[1][0]; ln_n[2 * pr + 1] = rw[1][1];
            }
        };
        // WIDE: u and v of column block `in`, read when the block is processed
        auto lnf_cols = [&](int in) {
            if constexpr (LNF && WIDE && STG) mm_lds_read_uv1<PAR_V>(mm_lds_addr(lds + PARAM) + (grp * HC + (uint32_t)in * 16u + q * 4u) * 4u, ln_u4[in], ln_v4[in]);
        };
        // the four values of accumulator (in, im) as they are rounded and stored: acc + bias, or the LayerNorm-folded form
        auto vals4 = [&](int in, int im, const float (&b)[4], float (&o)[4]) {
            const mm_f4& a = acc[in][im];
            if constexpr (LNF) {
                if constexpr (!SWAP) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j], ln_r[im], fmaf(ln_n[im], ln_u4[in][j], ln_v4[in][j]));
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = fmaf(a[j], ln_r[4 * im + j], fmaf(ln_n[4 * im + j], ln_u4[in][0], ln_v4[in][0]));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = a[j] + b[j];
            }
        };
        auto pack4 = [&](int in, int im, const float (&b)[4], uint32_t& lo, uint32_t& hi) {
            if constexpr (LNF) {
                float o[4];
                vals4(in, im, b, o);
                lo = MM<T>::cvt2(o[0], o[1]);
                hi = MM<T>::cvt2(o[2], o[3]);
            } else {
                const mm_f4& a = acc[in][im];
                lo = MM<T>::cvt2(a[0] + b[0], a[1] + b[1]);
                hi = MM<T>::cvt2(a[2] + b[2], a[3] + b[3]);
            }
        };
        // bias of this lane's own accumulator columns, all five column blocks up front (one wait, not one per block)
        uint2 bq_raw[5];
        if constexpr (STG && !LNF) {
#pragma unroll
            for (int in = 0; in < 5; ++in) bq_raw[in] = make_uint2(0, 0);
            if (bias) {                                        // staged behind the ring by wave 7 (dma_params)
                if (!stg) {                                    // (short tiles: now)
                    MM_BARRIER();
                    if (wave == 7u) { dma_params(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                    MM_BARRIER();
                }
                mm_lds_read5_b64(mm_lds_addr(lds + PARAM) + (grp * HC + q * 4u) * 2u, bq_raw);
            }
        } else
#pragma unroll
        for (int in = 0; in < 5; ++in) {
            bq_raw[in] = make_uint2(0, 0);
            const int32_t nb = n0 + (int32_t)(grp * HC + (uint32_t)in * 16u + q * 4u);       // N % 8 == 0: a 4-column group is inside N or outside
            if (!LNF && !transposed && bias && nb < p.N) bq_raw[in] = *reinterpret_cast<const uint2*>(bias + nb);
        }
        auto bias4 = [&](int in, float (&bq)[4]) {
            bq[0] = MM<T>::up((uint16_t)bq_raw[in].x); bq[1] = MM<T>::up((uint16_t)(bq_raw[in].x >> 16));
            bq[2] = MM<T>::up((uint16_t)bq_raw[in].y); bq[3] = MM<T>::up((uint16_t)(bq_raw[in].y >> 16));
        };
        auto swap16 = [&](uint32_t& a, uint32_t& b) {         // a's odd 16-lane rows <-> b's even rows
            const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
            a = r[0]; b = r[1];
        };
        auto swap32 = [&](uint32_t& a, uint32_t& b) {         // a's lanes 32..63 <-> b's lanes 0..31
            const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
            a = r[0]; b = r[1];
        };
        if (EPI == 0 || (QKV && !vtile)) {
            // dense rows: out[m, n] (+ resid[m, n]); row pointers hoisted, ten 16-byte stores per lane at immediate column offsets
            uint16_t* yrow[NPR];
            const uint16_t* rrow[NPR];
            bool live[NPR];
            const int32_t colb = n0 + (int32_t)(grp * HC + (q >> 1) * 8u);
            auto row_setup = [&](int pr) {
                const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q & 1u)) * 16u + li);
                live[pr] = m < p.M;
                const int64_t mm = live[pr] ? m : 0;
                yrow[pr] = Y + mm * p.ldy + colb;
                rrow[pr] = resid ? resid + mm * p.ldr + colb : nullptr;
            };
            // residual chunks are fetched RSD column blocks ahead of their use: all five up front on the 8-wave variant, two on the 12-wave one
            // (168 registers: 80 accumulators + the next tile's 36 fragment registers are live here)
            const bool rstat = EPI == 0 && p.rowstats != nullptr;
            float rs_s[NPR], rs_q[NPR];
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) { rs_s[pr] = __uint_as_float(zero_hi); rs_q[pr] = __uint_as_float(zero_hi); }
            constexpr int RSD = SPLIT ? 2 : (WIDE ? 2 : 5);
            uint4 rs[5][NPR];
            auto load_rs1 = [&](int in, int pr) {
                rs[in][pr] = make_uint4(zero_hi, zero_hi, zero_hi, zero_hi);
                if (colb + in * 16 < p.N) rs[in][pr] = *reinterpret_cast<const uint4*>(rrow[pr] + in * 16);
            };
            auto load_rs = [&](int in) {
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) load_rs1(in, pr);
            };
            auto block = [&](int in, int pr, const float (&bq)[4]) {
                uint32_t a0, a1, b0, b1;
                pack4(in, 2 * pr, bq, a0, a1);
                pack4(in, 2 * pr + 1, bq, b0, b1);
                swap16(a0, b0);
                swap16(a1, b1);
                uint32_t w4[4] = {a0, a1, b0, b1};
                if (resid) {
                    const uint32_t rsw[4] = {rs[in][pr].x, rs[in][pr].y, rs[in][pr].z, rs[in][pr].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) w4[k] = MM<T>::add2(w4[k], rsw[k]);           // one rounding of the exact sum, as before
                }
                if constexpr (WIDE) { if (in == 0 && pr == 0) { MM_STAMP(7); if (!resid) __builtin_amdgcn_s_waitcnt(0x0F70); MM_STAMP(8); } }      // vmcnt(0), as a BUILTIN (hipcc's waitcnt pass must see it: see the main loop): the next tile's stage 1 has landed
                if (live[pr] && colb + in * 16 < p.N) {
                    *reinterpret_cast<uint4*>(yrow[pr] + in * 16) = make_uint4(w4[0], w4[1], w4[2], w4[3]);      // the last N tile may be partial
                    if (rstat) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) MM<T>::stat2(w4[k], rs_s[pr], rs_q[pr]);
                    }
                }
            };
            // Row statistics for the LayerNorm that consumes this output (p.rowstats): (sum, sum of squares) of the stored values of each row over this
            // wave's 80 columns; one v_permlane32_swap + add folds the two 8-column halves of a lane pair, lanes < 32 then hold sums, lanes >= 32 squares
            auto flush_stat = [&](int pr) {
                uint32_t a = __float_as_uint(rs_s[pr]), b = __float_as_uint(rs_q[pr]);
                swap32(a, b);
                const float t = __uint_as_float(a) + __uint_as_float(b);
                const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q & 1u)) * 16u + li);
                if (m < p.M) p.rowstats[((int64_t)m * (NG * p.tiles_n) + NG * tile_n + (int32_t)grp) * 2 + zx(lane >> 5)] = t;
            };
            if constexpr (WIDE) {
                // Row pair by row pair, STRAIGHT-LINE: the wide tile only takes launches with M % 256 == 0 and N % 320 == 0 (gsw_mm_launch), so no lane and no column
                // block is ever masked, and the two launch-wide switches (residual operand, row records) select one of four copies of the body up front.  That matters
                // for more than the branches: behind ANY branch that contains a store hipcc's waitcnt pass gives up counting and waits vmcnt(0) for the next loaded
                // value -- with loads and stores in gfx9's ONE in-order counter that is a wait for every store issued so far, a full write round trip per block,
                // twenty per wave: the epilogue was 33-64 % of a tile on the residual launches (profiles/r05o_mm_trace_wide_*.txt).
                // The residual operand is a QUEUE of RD chunk loads in flight across the 20 (row pair, column block) blocks of the wave: the counted wait for block b's
                // chunk leaves the RD younger loads and the stores of the last RD blocks outstanding.  Registers: the main loop's fragment sets are dead here (the next
                // tile's first fragments are read at the top of the tile loop, not in front of the epilogue), and every block retires 8 accumulator registers.
                // Addresses: buffer descriptors, ONE lane offset per tensor for all 20 blocks, the block's offset in a scalar register.
                auto body = [&](auto res_tag, auto stat_tag) {
                    constexpr bool RES = decltype(res_tag)::value, STAT = decltype(stat_tag)::value;
                    constexpr int RD = LNF ? 10 : 8;          // (the plain epilogue holds the bias of all five column blocks: ten registers)
                    const uint32_t lrow = (q & 1u) * 16u + li, lcol = grp * (uint32_t)HC + (q >> 1) * 8u;
                    const uint32_t y_voff = (lrow * (uint32_t)p.ldy + lcol) * 2u;
                    const uint32_t y_s0 = (uint32_t)(((int64_t)(m0 + (int32_t)(wm * (16u * MT))) * p.ldy + n0) * 2);
                    mm_u4 rq[4 * 5];
                    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(RES ? resid : Y), 0, -1, 0x00020000);
                    uint32_t r_voff = 0u, r_s0 = 0u;
                    auto rq_issue = [&](int b) {
                        const int pr = b / 5, in = b % 5;
                        rq[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_q, (int)r_voff, (int)(r_s0 + (uint32_t)(pr * 32) * (uint32_t)p.ldr * 2u + (uint32_t)(in * 32)), 0);
                    };
                    if constexpr (RES) {
                        r_voff = (lrow * (uint32_t)p.ldr + lcol) * 2u;
                        r_s0 = (uint32_t)(((int64_t)(m0 + (int32_t)(wm * (16u * MT))) * p.ldr + n0) * 2);
#pragma unroll
                        for (int b = 0; b < RD; ++b) rq_issue(b);
                    }
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) {
                        lnf_rows(pr);
                        float ss = __uint_as_float(zero_hi), sq = __uint_as_float(zero_hi);
#pragma unroll
                        for (int in = 0; in < 5; ++in) {
                            const int b = pr * 5 + in;
                            float bq[4];
                            bias4(in, bq);
                            lnf_cols(in);
                            if constexpr (RES) { if (b + RD < 20) rq_issue(b + RD); }
                            uint32_t a0, a1, b0, b1;
                            pack4(in, 2 * pr, bq, a0, a1);
                            pack4(in, 2 * pr + 1, bq, b0, b1);
                            swap16(a0, b0);
                            swap16(a1, b1);
                            mm_u4 w4 = {a0, a1, b0, b1};
                            if constexpr (RES) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) w4[k] = MM<T>::add2(w4[k], rq[b][k]);          // one rounding of the exact sum, as before
                            }
                            if (b == 0) {
                                MM_STAMP(7);
                                // vmcnt(0), as a BUILTIN (hipcc's waitcnt pass must see it: see the main loop): the next tile's stage 1 has landed.  (With a residual
                                // operand the counted wait for chunk 0 -- in order behind those pieces -- has already said so.)
                                if constexpr (!RES) __builtin_amdgcn_s_waitcnt(0x0F70);
                                MM_STAMP(8);
                            }
                            // (a GLOBAL store -- scalar base + this lane's 32-bit offset -- not a buffer store: on gfx950 a VALU write to the data registers in the
                            // slot right behind a buffer_store_dwordx4 lands in the stored data even when soffset is an SGPR, the case hipcc's hazard recognizer
                            // does not pad: lanes 12-15 mod 16 of the second dword carried the next block's fp32 sums; tools/ubench/buffer_store_hazard.hip)
                            *reinterpret_cast<mm_u4*>(reinterpret_cast<uint8_t*>(Y) + (size_t)(y_s0 + (uint32_t)(pr * 32) * (uint32_t)p.ldy * 2u + (uint32_t)(in * 32)) + (size_t)y_voff) = w4;
                            if constexpr (STAT) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) MM<T>::stat2(w4[k], ss, sq);
                            }
                            __builtin_amdgcn_sched_barrier(0);                 // (the queue's depth is what the registers allow: no load moves up past a block)
                        }
                        if constexpr (STAT) {
                            // (sum, sum of squares) of the stored values of each row over this wave's 80 columns: see flush_stat
                            uint32_t a = __float_as_uint(ss), bb = __float_as_uint(sq);
                            swap32(a, bb);
                            const float t = __uint_as_float(a) + __uint_as_float(bb);
                            const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q & 1u)) * 16u + li);
                            p.rowstats[((int64_t)m * (NG * p.tiles_n) + NG * tile_n + (int32_t)grp) * 2 + zx(lane >> 5)] = t;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                if (resid) { if (rstat) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
                else { if (rstat) body(std::false_type{}, std::true_type{}); else body(std::false_type{}, std::false_type{}); }
            } else {
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) row_setup(pr);
                if (resid) {
#pragma unroll
                    for (int in = 0; in < RSD; ++in) load_rs(in);
                }
#pragma unroll
                for (int in = 0; in < 5; ++in) {
                    float bq[4];
                    bias4(in, bq);
                    if (resid && in + RSD < 5) load_rs(in + RSD);
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) block(in, pr, bq);
                }
                if (rstat) {
#pragma unroll
                    for (int pr = 0; pr < NPR; ++pr) flush_stat(pr);
                }
            }
        } else if (EPI == 1) {
            // per row pair p: this lane's output row and its addressing
            using orow_t = std::conditional_t<WIDE, int32_t, int64_t>;      // (WIDE: row indices fit 32 bits -- M does -- and four registers matter there)
            orow_t orow[NPR];
            int32_t img_b[NPR];
            bool live[NPR], border[NPR];
#pragma unroll
            for (int pr = 0; pr < NPR; ++pr) {
                const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q & 1u)) * 16u + li);
                orow[pr] = (orow_t)out_row(m, img_b[pr], border[pr], live[pr]);
            }
            // Column statistics for the GroupNorm that consumes this output (p.colstats): every lane adds the values it STORES -- as packed column pairs,
            // two v_dot2_f32_f16 per pair -- into 4 (sum, sum of squares) pairs; one v_permlane16_swap + add folds the two row tiles and leaves the sums
            // in even 16-lane rows and the sums of squares in odd ones, four row_shr steps fold the 16 rows of a lane row -- a fixed order, so the result
            // is reproducible -- and lanes 15 / 31 / 47 / 63 each write one 16-byte record (4 column pairs of one plane) for this wave's block of 16 MT rows.
            const bool cstat = p.colstats != nullptr;
            MM_STAMP(14);
            // WIDE: the row-bias / residual chunks of block (in, pr) are fetched PD blocks AHEAD (PD + 1 register sets): 20 blocks per wave, each a dependent
            // load -> add -> store chain otherwise, with the matrix pipe idle -- and, loads and stores sharing gfx9's in-order vmcnt, a chunk fetched one block
            // ahead still waits for the previous block's store to retire (see the dense-row epilogue: the queue there is ten deep)
            // (ONE prefetched operand: the residual when there is one, else the row bias -- no launch of the eps model or the VAE has both; a launch that
            // does fetches its row bias in place)
            constexpr int PD = WIDE ? 6 : 1;
            uint4 prev[PD + 1];
            auto fetch = [&](int in, int pr, int sl) {
                const int64_t col = (int64_t)n0 + grp * (uint32_t)HC + (uint32_t)in * 16u + (q >> 1) * 8u;
                // UNCONDITIONAL (lanes without a block read element 0, always there) and unconditionally consumed below: a load some path never uses is one
                // hipcc must assume in flight when its register is written next -- it then waits vmcnt(0), i.e. for this epilogue's stores, in the next tile's first phase
                const bool okl = live[pr] && col < p.N && !border[pr];
                if (resid) prev[sl] = *reinterpret_cast<const uint4*>(resid + (okl ? (int64_t)orow[pr] * p.ldr + col : (int64_t)0));
                else prev[sl] = *reinterpret_cast<const uint4*>(rowbias + (okl ? (int64_t)img_b[pr] * p.ldrb + col : (int64_t)0));
            };
            if constexpr (WIDE) {
                if (rowbias || resid) {
#pragma unroll
                    for (int j0 = 0; j0 < PD; ++j0) fetch(j0 / NPR, j0 % NPR, j0);
                }
            }
#pragma unroll
            for (int in = 0; in < 5; ++in) {
                float bq[4];
                bias4(in, bq);
                const int64_t col = (int64_t)n0 + grp * (uint32_t)HC + (uint32_t)in * 16u + (q >> 1) * 8u;
                float cs_s[4] = {0.f, 0.f, 0.f, 0.f}, cs_q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    constexpr int NB = 5 * NPR;
                    const int j = in * NPR + pr;
                    if constexpr (WIDE) { if ((rowbias || resid) && j + PD < NB) fetch((j + PD) / NPR, (j + PD) % NPR, (j + PD) % (PD + 1)); }
                    uint32_t a0, a1, b0, b1;
                    pack4(in, 2 * pr, bq, a0, a1);
                    pack4(in, 2 * pr + 1, bq, b0, b1);
                    swap16(a0, b0);
                    swap16(a1, b1);
                    if constexpr (WIDE) { if (j == 0 && !(rowbias || resid)) __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0), as a BUILTIN (hipcc's waitcnt pass must see it: see the main loop): the next tile's stage 1 has landed
                    if constexpr (WIDE) {
                        // the same block with every load consumed on every path (see fetch): arithmetic for all lanes, store and statistics for the live ones
                        uint32_t w4[4] = {a0, a1, b0, b1};
                        if (rowbias || resid) {
                            uint4 rb = make_uint4(0, 0, 0, 0), rs = make_uint4(0, 0, 0, 0);
                            if (resid) { rs = prev[j % (PD + 1)]; if (rowbias) rb = *reinterpret_cast<const uint4*>(rowbias + (int64_t)img_b[pr] * p.ldrb + (col < p.N ? col : 0)); }
                            else rb = prev[j % (PD + 1)];
                            const uint32_t rbw[4] = {rb.x, rb.y, rb.z, rb.w}, rsw[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if (rowbias) w4[k] = MM<T>::add2(w4[k], rbw[k]);
                                if (resid) w4[k] = MM<T>::add2(w4[k], rsw[k]);
                            }
                        }
                        if (border[pr]) { w4[0] = w4[1] = w4[2] = w4[3] = 0u; }
                        if (live[pr] && col < p.N) {
                            *reinterpret_cast<uint4*>(Y + (int64_t)orow[pr] * p.ldy + col) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                            if (cstat && !border[pr]) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) MM<T>::stat2(w4[k], cs_s[k], cs_q[k]);
                            }
                        }
                        continue;
                    }
                    if (!live[pr] || col >= p.N) continue;
                    uint32_t w4[4] = {a0, a1, b0, b1};
                    if (border[pr]) { w4[0] = w4[1] = w4[2] = w4[3] = 0u; }
                    else if (rowbias || resid) {
                        uint4 rb = make_uint4(0, 0, 0, 0), rs = make_uint4(0, 0, 0, 0);
                        if (rowbias) rb = *reinterpret_cast<const uint4*>(rowbias + (int64_t)img_b[pr] * p.ldrb + col);
                        if (resid) rs = *reinterpret_cast<const uint4*>(resid + (int64_t)orow[pr] * p.ldr + col);
                        const uint32_t rbw[4] = {rb.x, rb.y, rb.z, rb.w}, rsw[4] = {rs.x, rs.y, rs.z, rs.w};
                        // packed adds, rounded like the separate tensor adds they replace (conv1 has the row bias, conv2 / the token scatter the residual)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (rowbias) w4[k] = MM<T>::add2(w4[k], rbw[k]);
                            if (resid) w4[k] = MM<T>::add2(w4[k], rsw[k]);
                        }
                    }
                    *reinterpret_cast<uint4*>(Y + (int64_t)orow[pr] * p.ldy + col) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    if (cstat && !border[pr]) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) MM<T>::stat2(w4[k], cs_s[k], cs_q[k]);
                    }
                }
                if (cstat) {
                    float rec[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        uint32_t a = __float_as_uint(cs_s[k]), b = __float_as_uint(cs_q[k]);
                        swap16(a, b);                                                // a: (sum r0, sq r0, sum r2, sq r2), b: (sum r1, sq r1, sum r3, sq r3)
                        float t = __uint_as_float(a) + __uint_as_float(b);           // rows 0 / 2: sums of both row tiles, rows 1 / 3: their sums of squares
                        t = mm_dpp_add<0x111, 0xf>(t);                               // row_shr:1, 2, 4, 8: lane 15 of every 16-lane row = the row's total
                        t = mm_dpp_add<0x112, 0xf>(t);
                        t = mm_dpp_add<0x114, 0xf>(t);
                        t = mm_dpp_add<0x118, 0xf>(t);
                        rec[k] = t;
                    }
                    if (li == 15u && col < p.N) {
                        // records [block][plane: sums | sums of squares][N / 2 column pairs]
                        float* dst = p.colstats + (((int64_t)tile_m * WM + wm) * 2 + (q & 1u)) * (int64_t)(p.N >> 1) + (col >> 1);
                        *reinterpret_cast<mm_f4*>(dst) = mm_f4{rec[0], rec[1], rec[2], rec[3]};
                    }
                }
                MM_STAMP(8 + in);
            }
            // (WIDE: both prefetch registers are read once more HERE, so that on every path hipcc has placed its wait for their loads inside the epilogue -- a load
            // it must assume in flight when the register is written next costs a vmcnt(0), i.e. a wait for this epilogue's stores, in the next tile's first phase)
            if constexpr (WIDE) {
                if (rowbias || resid) {
#pragma unroll
                    for (int k = 0; k <= PD; ++k) asm volatile("" :: "v"(prev[k].x));
                }
            }
        } else if (EPI == 2) {
            // GEGLU: accumulator columns of an n-tile are [8 value | 8 gate] of outputs 8 in .. 8 in + 7 (within the group's 40 outputs): lanes < 32
            // hold values, lanes >= 32 the gates of the same four outputs 4 qv .. 4 qv + 3.  The projection is rounded to the storage dtype as torch
            // materialises it -- two at a time by v_cvt_pk, which is also the packing: ONE v_permlane32_swap of the packed pairs then leaves every lane
            // with (values, gates) of two outputs, the gate pair goes through gelu in fp32, is rounded like torch's F.gelu output, and the product is
            // one packed multiply (v_pk_mul_f16: the correctly rounded product of the two rounded operands).  33 instead of 52 instructions per accumulator.
            const uint32_t qv = q & 1u;
            uint32_t D[5][NPR][2];                            // [in][row pair][2 registers]: 4 consecutive outputs 4 qv .. of row tile 2p + (lane >> 5)
            auto gate_block = [&](int in) {                   // value * gelu(gate) of column block `in`, all row tiles of the wave
                float bq[4];
                bias4(in, bq);
                lnf_cols(in);
                uint32_t Wv[MT];                              // per row tile im: outputs 4 qv + 2 (lane >> 5) + {0, 1}, packed
#pragma unroll
                for (int im = 0; im < MT; ++im) {
                    uint32_t A, Bp;
                    pack4(in, im, bq, A, Bp);
                    swap32(A, Bp);                            // lanes < 32: A = values 0, 1, Bp = gates 0, 1; lanes >= 32: values 2, 3 and gates 2, 3
                    const uint32_t G = MM<T>::cvt2(mm_gelu(MM<T>::up_lo(Bp)), mm_gelu(MM<T>::up_hi(Bp)));
                    Wv[im] = MM<T>::mul2(A, G);
                }
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    uint32_t a = Wv[2 * pr], b = Wv[2 * pr + 1];
                    swap32(a, b);                             // lanes < 32: outputs (0, 1 | 2, 3) of row tile 2p; lanes >= 32: of row tile 2p + 1
                    D[in][pr][0] = a;
                    D[in][pr][1] = b;
                }
            };
            const int64_t obase = (int64_t)tile_n * (BN / 2) + grp * 40u;
            // n-tile pairs (0,1) and (2,3): one more swap round -> 8 consecutive outputs (16 bytes) per lane
            auto store_pair = [&](int ip) {
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    uint32_t a0 = D[2 * ip][pr][0], a1 = D[2 * ip][pr][1], b0 = D[2 * ip + 1][pr][0], b1 = D[2 * ip + 1][pr][1];
                    swap16(a0, b0);
                    swap16(a1, b1);
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q >> 1)) * 16u + li);
                    if (m < p.M)
                        *reinterpret_cast<uint4*>(Y + (int64_t)m * p.ldy + obase + zx((uint32_t)(2 * ip + (int)(q & 1u)) * 8u)) = make_uint4(a0, a1, b0, b1);
                }
            };
            // n-tile 4 has no partner: 8-byte stores (4 outputs per lane)
            auto store_last = [&]() {
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(lane >> 5)) * 16u + li);
                    if (m < p.M)
                        *reinterpret_cast<uint2*>(Y + (int64_t)m * p.ldy + obase + zx(32u + qv * 4u)) = make_uint2(D[4][pr][0], D[4][pr][1]);
                }
            };
            if constexpr (WIDE) {
                // a pair of column blocks at a time (2 x 4 row pairs x 2 registers live instead of 5 x 4 x 2 beside the accumulators still to come)
                gate_block(0); gate_block(1);
                MM_STAMP(7);
                __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0), as a builtin: the next tile's stage 1 has landed
                MM_STAMP(8);
                store_pair(0);
                __builtin_amdgcn_sched_barrier(0);
                gate_block(2); gate_block(3); store_pair(1);
                __builtin_amdgcn_sched_barrier(0);
                gate_block(4); store_last();
            } else {
#pragma unroll
                for (int in = 0; in < 5; ++in) gate_block(in);
                store_pair(0); store_pair(1);
                store_last();
            }
        } else {
            // transposed output Y[image][n][token]: lane (q, i) holds tokens 16 im + 4 q + j of output row 16 in + i
            // (QKV: the value part -- output rows counted from column n_rows, written to y2)
            const int32_t nbase = QKV ? p.n_rows : 0, Nt = p.N - nbase;
            uint16_t* Yt = QKV ? reinterpret_cast<uint16_t*>(p.y2) : Y;
#pragma unroll
            for (int in = 0; in < 5; ++in) {
                const int32_t ncol = n0 + (int32_t)(grp * HC + (uint32_t)in * 16u + li);
                const int32_t nrow = ncol - nbase;
                const bool nok = ncol < p.N;
                const float bv = !LNF && bias && nok ? MM<T>::up(bias[ncol]) : 0.f;
                const float bq[4] = {bv, bv, bv, bv};
#pragma unroll
                for (int pr = 0; pr < NPR; ++pr) {
                    uint32_t a0, a1, b0, b1;
                    pack4(in, 2 * pr, bq, a0, a1);
                    pack4(in, 2 * pr + 1, bq, b0, b1);
                    swap16(a0, b0);
                    swap16(a1, b1);
                    const int32_t m = m0 + (int32_t)(wm * (16u * MT) + (uint32_t)(2 * pr + (int)(q & 1u)) * 16u + (q >> 1) * 8u);      // first of 8 consecutive tokens
                    if (m >= p.M || !nok) continue;
                    const int32_t b = m / p.S, sidx = m - b * p.S;
                    *reinterpret_cast<uint4*>(Yt + ((int64_t)b * Nt + nrow) * p.S + sidx) = make_uint4(a0, a1, b0, b1);
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) acc[a][b] = mm_f4{0.f, 0.f, 0.f, 0.f};
        ++c_it;
    };

    auto mfma20 = [&](auto swap_tag, frag (&xc)[MT], frag (&wc)[5]) {
        constexpr bool SW = decltype(swap_tag)::value;
#pragma unroll
        for (int in = 0; in < 5; ++in)
#pragma unroll
            for (int im = 0; im < MT; ++im)
                acc[in][im] = SW ? MM_ABL_MMA(xc[im], wc[in], acc[in][im]) : MM_ABL_MMA(wc[in], xc[im], acc[in][im]);
    };
    // issue order of a phase's main block: three DMA pieces, each behind three MFMAs, then the nine fragment reads one per MFMA
    auto pin_order = [&]() {
        // (fragment reads in FRONT of the DMA pieces / at one per MFMA from the start of the phase was measured 3-9 % slower in both variants)
        constexpr int NM = 5 * MT, NR = MT + 5;              // MFMAs and fragment reads of a phase: 20 / 9, or 10 / 7
        if constexpr (SPLIT) {
            constexpr int per = MT == 4 ? 2 : 1;             // MFMAs in front of each fragment read
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - per * NR, 0);
        } else {
            constexpr int NV = NPA / 2 + 1;                  // DMA pieces of a phase (without the extra one): 3, or 2
            constexpr int per = MT == 4 ? 3 : 1;             // MFMAs in front of each DMA piece
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);    // one LDS-DMA piece
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // one fragment read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NM - per * NV - NR, 0);
        }
    };
    // X holds the fragments of (stage s, k-half 0) on entry; Y is filled with (s, k-half 1), then X with (s+1, k-half 0)
    auto step = [&](auto swap_tag, frag (&xX)[MT], frag (&wX)[5], frag (&xY)[MT], frag (&wY)[5], const int32_t step_i) {
        const uint32_t nx_slot = rd_slot + STAGE == RING ? 0u : rd_slot + STAGE;
        // ---- even phase: second half of stage s+2 (its slot held stage s-1, whose last reads completed before the previous barrier)
        MM_STAMP(0);
        if constexpr (!SPLIT) { dma_extra(); dma_h2(); }
        MM_ABL_READ(read_frags(xY, wY, rd_slot, 1u));
        mfma20(swap_tag, xX, wX);
        pin_order();
        if constexpr (!SPLIT) { if (--pr_run == 0) end_run(); }
        MM_STAMP(1);
        // stage s+1 is read in the odd phase: everything but the newest stage (s+2, both halves) must have landed
        if constexpr (STG) {
            // wave 7, step 1: its parameter pieces of step 0 may stay in flight too
            if (stg7 && step_i == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA + NPAR) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        } else if constexpr (!SPLIT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
        MM_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MM_BARRIER();
        MM_STAMP(3);
        // ---- odd phase: first half of stage s+3, into the slot of stage s (its last reads completed before the barrier above)
        if constexpr (STG) {
            // every wave is past the previous tile's epilogue (its parameter reads were drained in front of the barrier above)
            if (stg7 && step_i == 0) dma_params();
        }
        if constexpr (!SPLIT) dma_h1();
        MM_ABL_READ(read_frags(xX, wX, nx_slot, 0u));
        mfma20(swap_tag, xY, wY);
        pin_order();
        MM_STAMP(4);
        // no lgkmcnt(0) either: the compiler's own counted waits let the next phase's first MFMAs start as soon as THEIR fragments are in,
        // and the slot these reads come from is only released by the next barrier, which drains them
        // no barrier here: the next even phase only writes the slot freed by the barrier above and only reads the stage that barrier
        // published, so the waves of a SIMD are free to drift by up to one phase -- one's DMA / fragment reads under the other's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        MM_STAMP(5);
        rd_slot = nx_slot;
    };

    if constexpr (WIDE) {
        // ------------------------------------------------------------ the 256 x 320 tile: a wave multiplies 128 x 80 outputs, 40 MFMAs per phase.
        // Registers: 160 accumulators, so fragments cannot be held twice.  W fragments (5 per phase) are double-buffered across phases; the 8 activation
        // blocks of a phase stream through a ring of four register sets, block im + 3 read while block im multiplies (the last three reads of a phase fetch
        // blocks 0..2 of the next one).  Ring of TWO stage slots, ONE barrier per stage, inside the odd phase behind block 4: by then every wave has read the
        // last fragment of stage s (block 7 of k-half 1 is fetched in group 4) and waited for its own pieces of stage s + 1, so the barrier both publishes
        // stage s + 1 (read from group 5 on) and frees the slot of stage s.
        // Where the DMA goes: the nine pieces of stage s + 2, into that freed slot, one per group over the TAIL of the odd phase (groups 5..7) and the first six
        // groups of the following even phase -- a piece holds its wave at issue for 60-185 cycles, three per group (measured: all nine in the tail) cost the long-K
        // convolutions half of what the tile gains.  Only the tail in front of an EPILOGUE issues all nine (stage 1 of the next tile; once per tile).
        // The step sequence is rotated around the barrier -- per tile: [even 0 | odd head 0 | barrier] { odd tail s + DMA | even s + 1 | odd head s + 1 |
        // vmcnt(0) | barrier } [odd tail P - 1 + DMA | epilogue] -- for three reasons that all come from gfx9's single vmcnt:
        //  * hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of the first fragment read behind a loop's back edge whenever an LDS-DMA may be in flight there:
        //    the back edge sits right behind the barrier, where this wave's vmcnt(0) has just drained everything, so that wait is free (in the first build, with
        //    pieces in flight across the back edge, it stalled every stage);
        //  * the DMA behind the last step of a tile is stage 1 of the NEXT tile, issued in front of the epilogue and waited for in front of the epilogue's first
        //    store; the first barrier of a tile (peeled out of the loop) therefore needs NO vmcnt wait.  A wait there is a wait for the epilogue's stores (stores
        //    count in vmcnt, and nothing orders a load's return against them but vmcnt(0)): with every CU leaving its epilogue at the same time that is a ~40 MB
        //    write burst, ~9 us per tile with the matrix pipe idle -- the intercept of time against K that the 12-wave form, whose multiplying waves never wait on
        //    vmcnt, does not have;
        //  * no step-dependent branch is left inside the loop (parameter staging, the first-step exception: all in the peeled part).
        // fragment addresses: ONE persistent lane offset; a segment derives the three bases it reads from (this phase's activation blocks, the next phase's
        // activation blocks and W set) behind an empty asm -- left to itself hipcc hoists all eight (slot, k-half, operand) combinations into registers
        auto rd_at = [&](uint32_t base, int blk) -> frag { return *reinterpret_cast<const frag*>(lds + base + blk * 2048); };
        frag wA[5], wB[5], xr[4];
        // groups IM0 .. IM1 - 1 of a phase (8 groups of 5 MFMAs: one activation block against the phase's five W fragments)
        // DMA: 0 = this segment issues nothing, 1 = the steady state (tail: pieces 0..2, even phase: pieces 3..8), 2 = the tail in front of an epilogue (all nine)
        auto seg = [&](auto odd_tag, auto im0_tag, auto im1_tag, auto dma_tag, frag (&wc)[5], frag (&wn)[5], const uint32_t slot_c, const uint32_t slot_n) {
            constexpr bool ODD = decltype(odd_tag)::value;
            constexpr int IM0 = decltype(im0_tag)::value, IM1 = decltype(im1_tag)::value, DMA = decltype(dma_tag)::value;
            constexpr uint32_t kh_c = ODD ? 1u : 0u, kh_n = ODD ? 0u : 1u;
            uint32_t lrd = lane_rd0;
            asm volatile("" : "+v"(lrd));
            const uint32_t a_c = (lrd ^ (kh_c * 64u)) + (wm * (16u * MT) * 128u + slot_c);
            const uint32_t a_n = (lrd ^ (kh_n * 64u)) + (wm * (16u * MT) * 128u + slot_n);
            const uint32_t w_n = (lrd ^ (kh_n * 64u)) + ((uint32_t)BM * 128u + grp * (uint32_t)HC * 128u + slot_n);
#pragma unroll
            for (int im = IM0; im < IM1; ++im) {
                // activation block im + 3 of this phase, or block im - 5 of the next one, into the ring entry block im - 1 has left
                // (DMA == 2, the tail in front of an epilogue: the next phase is the next TILE's step 0 -- its fragments are read at the top of the tile loop instead:
                // read here they would sit in 32 registers across the whole epilogue, which needs them for its queue of residual loads)
                if (im + 3 < 8) { MM_ABL_READ(xr[(im + 3) & 3] = rd_at(a_c, im + 3)); }
                else if (DMA != 2) { MM_ABL_READ(xr[(im + 3) & 3] = rd_at(a_n, im - 5)); }
                if (DMA != 2 && im == 5) { MM_ABL_READ(wn[0] = rd_at(w_n, 0)); MM_ABL_READ(wn[1] = rd_at(w_n, 1)); MM_ABL_READ(wn[2] = rd_at(w_n, 2)); }
                if (DMA != 2 && im == 6) { MM_ABL_READ(wn[3] = rd_at(w_n, 3)); MM_ABL_READ(wn[4] = rd_at(w_n, 4)); }
                if (DMA == 2 && ODD && im >= 5) { dma_piece8(3 * (im - 5)); dma_piece8(3 * (im - 5) + 1); dma_piece8(3 * (im - 5) + 2); }
                if (DMA == 1 && ODD && im >= 5) dma_piece8(im - 5);
                if (DMA == 1 && !ODD && im < 6) dma_piece8(3 + im);
#pragma unroll
                for (int in = 0; in < 5; ++in) acc[in][im] = MM_ABL_MMA(wc[in], xr[im & 3], acc[in][im]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // the producer's slow path (next tap row / channel block / segment / tile), due when the stage just issued ended its run: right BEHIND a barrier, where one
        // fragment set is dead -- in the tail of the odd phase, beside both W sets, its temporaries spilled fragments
        auto run_end = [&]() { if (pr_end) { end_run8(); pr_end = false; } };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        using I5 = std::integral_constant<int, 5>; using I8 = std::integral_constant<int, 8>;
        for (uint32_t it = 0; it < nt_mine; ++it) {
            uint32_t nx_slot = rd_slot + STAGE == RING ? 0u : rd_slot + STAGE;
            MM_STAMP(6);                                                  // (MM_TRACE builds; interval 6: the epilogue of the previous tile)
            // the tile's first fragments (stage 0 landed before the previous epilogue's first store / in the prologue)
            {
                uint32_t lrd = lane_rd0;
                asm volatile("" : "+v"(lrd));
                const uint32_t a0_ = lrd + (wm * (16u * MT) * 128u + rd_slot), w0_ = lrd + ((uint32_t)BM * 128u + grp * (uint32_t)HC * 128u + rd_slot);
#pragma unroll
                for (int in = 0; in < 5; ++in) { MM_ABL_READ(wA[in] = rd_at(w0_, in)); }
#pragma unroll
                for (int im = 0; im < 3; ++im) { MM_ABL_READ(xr[im] = rd_at(a0_, im)); }
            }
            // step 0 up to its barrier: stage 1 landed long ago (prologue / top of the previous epilogue) -- no vmcnt wait
            seg(std::false_type{}, I0{}, I8{}, I0{}, wA, wB, rd_slot, rd_slot);
            seg(std::true_type{}, I0{}, I5{}, I0{}, wB, wA, rd_slot, nx_slot);
            MM_STAMP(0);                                                  // interval 0: step 0 up to its barrier (issue)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's last fragment reads of stage 0
            MM_BARRIER();
            MM_STAMP(1);                                                  // interval 1: the wait at the first barrier
            run_end();
            if constexpr (STG) { if (stg7) dma_params(); }                // every wave is past the previous tile's epilogue (its parameter reads): covered by the next vmcnt(0)
            for (int32_t i = 1; i < P_mine; ++i) {
                seg(std::true_type{}, I5{}, I8{}, I1{}, wB, wA, rd_slot, nx_slot);    // tail of step i - 1: next fragments + pieces 0..2 of stage i + 1
                rd_slot = nx_slot;
                nx_slot = rd_slot + STAGE == RING ? 0u : rd_slot + STAGE;
                seg(std::false_type{}, I0{}, I8{}, I1{}, wA, wB, rd_slot, rd_slot);   // + pieces 3..8
                seg(std::true_type{}, I0{}, I5{}, I0{}, wB, wA, rd_slot, nx_slot);
                MM_STAMP(2);                                              // interval 2: a steady-state step's issue (tail + even + odd head)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of stage i + 1 (issued a stage ago)
                MM_STAMP(3);                                              // interval 3: its vmcnt wait
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // its last fragment reads of stage i
                MM_BARRIER();
                MM_STAMP(4);                                              // interval 4: its barrier wait
                run_end();
            }
            seg(std::true_type{}, I5{}, I8{}, I2{}, wB, wA, rd_slot, nx_slot);        // tail of the last step: the next tile's first fragments + ALL of its stage 1
            MM_STAMP(5);                                                              // interval 5: that tail
            rd_slot = nx_slot;
            epilogue();                                                               // (waits for those pieces in front of its first store)
        }
        MM_STAMP(6);
        MM_TRACE_DUMP();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the filler DMA of the last steps
        return;
    }
    // The step loop is a loop of its own (not one flat loop with the epilogue inside): hipcc then places the wait for the epilogue's
    // loads / stores once in front of it instead of inside the steady state.
    frag xa[MT], wa[5], xb[MT], wb[5];
    read_frags(xa, wa, 0u, 0u);                               // (stage 0, k-half 0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (uint32_t it = 0; it < nt_mine; ++it) {
        if constexpr (QKV) {
            int32_t tm_, tn_;
            decode_tile(it * G + slotx, tm_, tn_);
            if (tn_ * BN >= p.n_rows) { for (int32_t i = 0; i < P_mine; ++i) step(std::true_type{}, xa, wa, xb, wb, i); }
            else { for (int32_t i = 0; i < P_mine; ++i) step(std::false_type{}, xa, wa, xb, wb, i); }
        } else {
            for (int32_t i = 0; i < P_mine; ++i) step(std::integral_constant<bool, SWAP>{}, xa, wa, xb, wb, i);
        }
        epilogue();
        MM_STAMP(6);
    }
    MM_TRACE_DUMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the filler DMA of the last steps
}


// Second half of a split-K launch: every thread owns one float4 of a tile's accumulator image (the layout the EPI 4 kernel dumps: [wave][5 * MT
// accumulators][lane]; lane (q = lane >> 4, i = lane & 15) holds columns 16 in + 4 q .. + 3 of row 16 im + i of the wave's 16 MT x 80 block), adds the
// `splits` slabs in split order (fixed order: deterministic) and runs the epilogue of the launch's mode with the rounding points of the fused epilogues.
template <typename T>
__global__ __launch_bounds__(256) void gsw_mm_reduce_kernel(const MMArgs p, const int MT) {
    const int32_t BM = 64 * MT, upt = 8 * 5 * MT * 64;         // float4 units per tile
    const int64_t total = (int64_t)p.ntiles * upt;
    const size_t split_stride = (size_t)total * 4;              // floats between the slabs of consecutive splits
    const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.bias);
    const uint16_t* rowbias = reinterpret_cast<const uint16_t*>(p.rowbias);
    const uint16_t* resid = reinterpret_cast<const uint16_t*>(p.resid);
    uint16_t* Y = reinterpret_cast<uint16_t*>(p.y);
    const bool compact = (p.flags & MM_FLAG_COMPACT) != 0;
    const int32_t HpWp = p.Hp * p.Wp;
    const uint32_t tiles_m = (uint32_t)p.ntiles / (uint32_t)p.tiles_n;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const uint32_t lane = (uint32_t)u & 63u;
        uint32_t r = (uint32_t)(u >> 6);
        const uint32_t accidx = r % (uint32_t)(5 * MT); r /= (uint32_t)(5 * MT);
        const uint32_t wave = r & 7u, L = r >> 3;
        const uint32_t in = accidx / (uint32_t)MT, im = accidx - in * (uint32_t)MT;
        const uint32_t wm = wave & 3u, grp = wave >> 2, q = lane >> 4, li = lane & 15u;
        // logical tile -> (tile_m, tile_n): the panel order of the engine
        const uint32_t pw = (uint32_t)p.panel, full = tiles_m * pw, pn = L / full, rem = L - pn * full;
        const uint32_t width = min(pw, (uint32_t)p.tiles_n - pn * pw);
        const int32_t tile_m = (int32_t)(rem / width), tile_n = (int32_t)(pn * pw + rem - (rem / width) * width);
        const int32_t m = tile_m * BM + (int32_t)(wm * (uint32_t)(16 * MT) + im * 16u + li);
        const int32_t n = tile_n * 160 + (int32_t)(grp * 80u + in * 16u + q * 4u);
        const bool geglu = p.mode == MM_MODE_GEGLU;
        if (m >= p.M || n >= p.N || (geglu && q >= 2u)) continue;
        mm_f4 a = mm_f4{0.f, 0.f, 0.f, 0.f}, g = mm_f4{0.f, 0.f, 0.f, 0.f};
        const float* src = p.ws + (size_t)u * 4;
        // the slabs of a tile are added in split order (a fixed order: deterministic), but LOADED eight at a time: one load per dependent add made this
        // kernel latency-bound on its split count (16 splits = 16 L2 round trips in a row)
        int s = 0;
        for (; s + 8 <= p.splits; s += 8) {
            mm_f4 t[8], tg[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = *reinterpret_cast<const mm_f4*>(src + (size_t)(s + j) * split_stride);
            if (geglu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tg[j] = *reinterpret_cast<const mm_f4*>(src + (size_t)(s + j) * split_stride + 32 * 4);       // the gate columns: lane + 32
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) a += t[j];
            if (geglu) {
#pragma unroll
                for (int j = 0; j < 8; ++j) g += tg[j];
            }
        }
        for (; s + 2 <= p.splits; s += 2) {
            const mm_f4 t0 = *reinterpret_cast<const mm_f4*>(src + (size_t)s * split_stride), t1 = *reinterpret_cast<const mm_f4*>(src + (size_t)(s + 1) * split_stride);
            mm_f4 g0 = mm_f4{0.f, 0.f, 0.f, 0.f}, g1 = g0;
            if (geglu) { g0 = *reinterpret_cast<const mm_f4*>(src + (size_t)s * split_stride + 32 * 4); g1 = *reinterpret_cast<const mm_f4*>(src + (size_t)(s + 1) * split_stride + 32 * 4); }
            a += t0; a += t1;
            if (geglu) { g += g0; g += g1; }
        }
        for (; s < p.splits; ++s) {
            a += *reinterpret_cast<const mm_f4*>(src + (size_t)s * split_stride);
            if (geglu) g += *reinterpret_cast<const mm_f4*>(src + (size_t)s * split_stride + 32 * 4);
        }
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) { for (int j = 0; j < 4; ++j) b4[j] = MM<T>::up(bias[n + j]); }
        uint16_t h[4];
        for (int j = 0; j < 4; ++j) h[j] = MM<T>::cvt(a[j] + b4[j]);
        if (geglu) {
            for (int j = 0; j < 4; ++j) {
                const float gb = bias ? MM<T>::up(bias[n + 8 + j]) : 0.f;
                const float gg = MM<T>::up(MM<T>::cvt(g[j] + gb));
                const float ge = MM<T>::up(MM<T>::cvt(mm_gelu(gg)));
                h[j] = MM<T>::cvt(MM<T>::up(h[j]) * ge);
            }
            const int64_t ocol = (int64_t)tile_n * 80 + grp * 40u + in * 8u + q * 4u;
            *reinterpret_cast<uint2*>(Y + (int64_t)m * p.ldy + ocol) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
            continue;
        }
        if (p.mode == MM_MODE_TRANS || (p.mode == MM_MODE_QKV && n >= p.n_rows)) {
            const int32_t nbase = p.mode == MM_MODE_QKV ? p.n_rows : 0;
            uint16_t* Yt = p.mode == MM_MODE_QKV ? reinterpret_cast<uint16_t*>(p.y2) : Y;
            const int32_t b = m / p.S, sidx = m - b * p.S;
            for (int j = 0; j < 4; ++j) Yt[((int64_t)b * (p.N - nbase) + (n - nbase) + j) * p.S + sidx] = h[j];
            continue;
        }
        int64_t orow = m;
        int32_t img = 0;
        bool border = false;
        if (p.mode == MM_MODE_PF || p.mode == MM_MODE_UP2X) {
            int32_t b, yy, xx;
            if (compact) {
                const int32_t Wi = p.Wp - 2, HW = (p.Hp - 2) * Wi;
                b = m / HW;
                const int32_t rr = m - b * HW;
                yy = rr / Wi; xx = rr - yy * Wi + 1; yy += 1;
                orow = (int64_t)b * HpWp + yy * p.Wp + xx;
            } else {
                b = m / HpWp;
                const int32_t rr = m - b * HpWp;
                yy = rr / p.Wp; xx = rr - yy * p.Wp;
                border = (yy == 0) | (yy == p.Hp - 1) | (xx == 0) | (xx == p.Wp - 1);
            }
            img = b;
            if (p.mode == MM_MODE_UP2X) {
                if (border) continue;
                const int32_t dy = (p.up - 1) >> 1, dx = (p.up - 1) & 1;
                orow = ((int64_t)b * (2 * p.Hp - 2) + (2 * (yy - 1) + dy + 1)) * (2 * p.Wp - 2) + (2 * (xx - 1) + dx + 1);
            }
        } else if (p.mode == MM_MODE_TOK2PF) {
            const int32_t b = m / p.S, ii = m - b * p.S;
            const int32_t yy = ii / p.Wimg, xx = ii - yy * p.Wimg;
            img = b;
            orow = (int64_t)b * HpWp + (int64_t)(yy + 1) * p.Wp + (xx + 1);
        }
        if (border) { h[0] = h[1] = h[2] = h[3] = 0; }
        else if (rowbias || resid) {
            for (int j = 0; j < 4; ++j) {
                float f = MM<T>::up(h[j]);
                if (p.mode == MM_MODE_DENSE && !rowbias) f += MM<T>::up(resid[orow * p.ldr + n + j]);       // EPI 0: one add
                else f = f + (rowbias ? MM<T>::up(rowbias[(int64_t)img * p.ldrb + n + j]) : 0.f) + (resid ? MM<T>::up(resid[orow * p.ldr + n + j]) : 0.f);
                h[j] = MM<T>::cvt(f);
            }
        }
        *reinterpret_cast<uint2*>(Y + orow * p.ldy + n) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    }
}

// host ---------------------------------------------------------------------------------------------
template <typename T, int EPI, bool SPLIT, int MT, bool LNF = false>
int mm_launch_k(const MMArgs& a, uint32_t grid, hipStream_t st) {
    static bool attr_done = false;          // benign race: setting the attribute twice is harmless
    constexpr bool wide = MT == 8;          // the 256 x 320 tile: two stage slots of 72 KiB
    constexpr size_t ldsb = wide ? 2u * 576u * 128u + 6144u + 256u          // two stage slots + the staged epilogue parameters + the landing area of the residual touches
                                 : 3u * (size_t)(64 * MT + 160) * 128u + (!SPLIT && (EPI == 2 || (EPI == 0 && !LNF)) ? 4096u : 0u);   // the ring (+ the staged epilogue parameters)
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)gsw_mm_kernel<T, EPI, SPLIT, MT, LNF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL((gsw_mm_kernel<T, EPI, SPLIT, MT, LNF>), dim3(grid), dim3(SPLIT ? 768 : 512), ldsb, st, a);
    return (int)hipGetLastError();
}
template <typename T, int EPI>
int mm_launch_t(const MMArgs& a, uint32_t grid, int mt, hipStream_t st) {
    if constexpr (EPI == 5) return mt == 4 ? mm_launch_k<T, 5, false, 4>(a, grid, st) : mm_launch_k<T, 5, false, 2>(a, grid, st);      // dense epilogues: 8-wave form
    else {
    if (a.ln_stat) {                      // LayerNorm folded into the epilogue: dense rows / GEGLU / transposed, 8-wave form
        if constexpr (EPI == 0 || EPI == 2) { if (mt == 8) return mm_launch_k<T, EPI, false, 8, true>(a, grid, st); }
        if constexpr (EPI == 0 || EPI == 2 || EPI == 3)
            return mt == 4 ? mm_launch_k<T, EPI, false, 4, true>(a, grid, st) : mm_launch_k<T, EPI, false, 2, true>(a, grid, st);
        else return (int)hipErrorInvalidValue;
    }
    if constexpr (EPI == 0 || EPI == 1 || EPI == 2) { if (mt == 8) return mm_launch_k<T, EPI, false, 8>(a, grid, st); }
    // bit e of the split mask set = epilogue kind e runs the 12-wave variant whose waves 8-11 own the LDS-DMA (gsw_mm_config / GSW_MM_SPLIT: A/B switch)
    const bool split = (g_mm_split_mask.load(std::memory_order_relaxed) >> EPI) & 1;
    // (dense rows, 256-row tile: the 12-wave form does not fit its 168 registers -- 32-40 bytes of scratch per lane -- so that combination is not
    // instantiated and always runs the 8-wave form; the mask bit still selects the 12-wave 128-row variant)
    if constexpr (EPI == 0) { if (mt == 4) return mm_launch_k<T, EPI, false, 4>(a, grid, st); }
    else { if (mt == 4) return split ? mm_launch_k<T, EPI, true, 4>(a, grid, st) : mm_launch_k<T, EPI, false, 4>(a, grid, st); }
    return split ? mm_launch_k<T, EPI, true, 2>(a, grid, st) : mm_launch_k<T, EPI, false, 2>(a, grid, st);
    }
}
template <typename T>
int mm_launch_splitk(const MMArgs& a, uint32_t grid, int mt, hipStream_t st) {
    // 128- or 256-row tiles (mm_plan decides); both variants fit the 168 registers of the 12-wave form (152 / 100) without scratch
    if (mt != 2 && mt != 4) return (int)hipErrorInvalidValue;
    const int e = mt == 4 ? mm_launch_k<T, 4, true, 4>(a, grid, st) : mm_launch_k<T, 4, true, 2>(a, grid, st);
    if (e != 0) return e;
    const int64_t units = (int64_t)a.ntiles * 8 * 5 * mt * 64;
    hipLaunchKernelGGL((gsw_mm_reduce_kernel<T>), dim3((uint32_t)std::min<int64_t>((units + 255) / 256, 2048)), dim3(256), 0, st, a, mt);
    return (int)hipGetLastError();
}
template <typename T>
int mm_launch_e(const MMArgs& a, int epi, uint32_t grid, int mt, hipStream_t st) {
    switch (epi) {
        case 0: return mm_launch_t<T, 0>(a, grid, mt, st);
        case 1: return mm_launch_t<T, 1>(a, grid, mt, st);
        case 2: return mm_launch_t<T, 2>(a, grid, mt, st);
        case 5: return mm_launch_t<T, 5>(a, grid, mt, st);
        default: return mm_launch_t<T, 3>(a, grid, mt, st);
    }
}

}  // namespace

extern __attribute__((visibility("hidden"))) thread_local int g_last_hip_error;   // gswm_kernels.hip


// An engine launch without extras (a nullptr `ex` of an entry point): no records requested, no split-K scratch.  ABI < 0.5.0 kept thread-local one-shot
// requests and a thread-local workspace behind this case; since 0.5.0 everything a launch needs travels in the caller's GswMmExtras.
void gsw_mm_no_extras(GswMmExtras* ex) {
    ex->colstats_dev = nullptr; ex->colstats_capacity = 0;
    ex->rowstats_dev = nullptr; ex->rowstats_capacity = 0;
    ex->workspace_dev = nullptr; ex->workspace_bytes = 0; ex->max_splits = 0;
    ex->colstats_rows_per_block = 0; ex->colstats_blocks = 0; ex->rowstats_slots = 0; ex->splits = 1; ex->flags = 0;
}

int gsw_build_flags(void) { return GSW_MM_BUILD_FLAGS; }

int gsw_mm_config(int tile_rows, int split_mask) {
    if (tile_rows != 0 && tile_rows != 128 && tile_rows != 256 && tile_rows != 512 && tile_rows != -1) return GSW_ERR_BAD_ARG;      // 512: the 256 x 320 tile wherever it is legal
    if (split_mask < -1 || split_mask > 15) return GSW_ERR_BAD_ARG;
    if (tile_rows != -1) g_mm_tile_rows.store(tile_rows, std::memory_order_relaxed);
    if (split_mask != -1) g_mm_split_mask.store(split_mask, std::memory_order_relaxed);
    return GSW_OK;
}

int gsw_mm_get_config(int* tile_rows, int* split_mask) {
    if (tile_rows) *tile_rows = g_mm_tile_rows.load(std::memory_order_relaxed);
    if (split_mask) *split_mask = g_mm_split_mask.load(std::memory_order_relaxed);
    return GSW_OK;
}

// Tiling and split-K plan of a launch, with the time the cost model predicts for it (microseconds).  The model is fitted to tools/splitk_tile_sweep.py
// (profiles/r04i_splitk_tile_sweep.txt: 35 shapes x tile rows x split counts, HBM-cold weights, graph-captured; rms error 6 %, mean regret of its choices 1 %):
//   a stage of a 128-row tile costs 0.56 us while at most half the CUs work and 0.67 with all of them; a stage of a 256-row tile 0.79-0.81 and 1.13 (the chip is
//   power- and L2-bound when every CU multiplies: section 4.8 of DESIGN.md); an unsplit launch pays 2.5 us on top, a split one 12.5 (its reduce kernel) and 0.014 per
//   80 KiB of slab.
// Unsplit tile: for long K (>= 40 stages) and at most one round of 256-row tiles the model decides -- 128-row tiles when 256-row ones would leave half the chip idle
// (4096 x 1280 outputs: 127 vs 162 us at K = 11520; 52 vs 69 at K = 5120; the round-3 rule kept 256 rows there from a sweep whose weights were L2-hot).  Short K keeps
// the measured rule of round 3 (128-row tiles whenever 256-row ones do not fill the chip: a short-K weight matrix stays in L2 and all CUs win), more than one round
// keeps 256-row tiles (a half tile re-fetches the weight tile twice as often).
// Split: at most 128 tiles, at least 8 stages, up to 32 ways (one image at 8 x 8: 8 tiles x 32 = the whole chip, 19.9 vs 21.1 us at 16 ways) and 256 workgroups, 128- or 256-row tiles, taken for a predicted gain of 5 % or more.  Forced splits
// (max_splits > 1: tests) use 128 rows unless gsw_mm_config forces the 256-row tile.
// Compute units the persistent grid, the plan's "rounds" and its half-chip threshold are counted in: 256, the MI355X the stage costs, the wide-tile thresholds and the 8-XCD
// tile interleave were fitted and tested on.  A partition with fewer CUs still runs correctly with 256 workgroups (they queue); GSW_MM_CUS=<n> (a multiple of 8) or
// GSW_MM_CUS=device opts into another count for experiments -- read once per process.
static int mm_cus() {
    static const int cus = [] {
        const char* e = getenv("GSW_MM_CUS");
        if (!e || !*e) return 256;
        int n = atoi(e);
        if (n <= 0) {                                   // "device"
            int dev = 0;
            hipDeviceProp_t pr;
            n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256;
        }
        return n >= 8 ? n / 8 * 8 : 256;
    }();
    return cus;
}
struct MMPlan { int bm; int splits; double t_us; };
// (the stage costs were fitted on ONE box of the pool in round 4 -- profiles/r04i_splitk_tile_sweep.txt: MI355X, 256 CUs, 1400 W board limit, HBM-cold weights,
// the deep levels at 4-64 images, board at 1.1-1.3 kW; boxes of the pool differ by +-4 %.  W = busy workgroups; the costs are functions of the busy FRACTION
// of the chip, so a different CU count rescales W, not the constants.)
static inline double mm_stage_us(int bm, double W, int64_t CU) {
    const double half = 0.5 * (double)CU;
    const double over = W > half ? (W - half) / half : 0.0;
    return bm == 128 ? 0.56 + 0.11 * over : 0.79 + 0.02 * std::min(1.0, W / half) + 0.32 * over;
}
static MMPlan mm_plan(int64_t M, int64_t tiles_n, int32_t P, bool can_split, int max_splits) {
    const int bm_env = g_mm_tile_rows.load(std::memory_order_relaxed);          // GSW_MM_BM / gsw_mm_config: 128 / 256 forces a tiling (A/B runs, tests)
    const int64_t CU = mm_cus();
    const int64_t nt256 = ((M + 255) / 256) * tiles_n, nt128 = ((M + 127) / 128) * tiles_n;
    auto t_unsplit = [&](int bm) {
        const int64_t nt = bm == 256 ? nt256 : nt128, rounds = (nt + CU - 1) / CU;
        return 2.5 + mm_stage_us(bm, rounds == 1 ? (double)nt : (double)CU, CU) * (double)P * (double)rounds;
    };
    int BM;
    if (bm_env == 128 || bm_env == 256) BM = bm_env;
    else if (M <= 128) BM = 256;
    else if (P >= 40 && nt256 <= CU) BM = t_unsplit(128) < t_unsplit(256) ? 128 : 256;
    else BM = nt256 < (P >= 64 ? CU / 2 : CU) ? 128 : 256;
    MMPlan pl{BM, 1, t_unsplit(BM)};
    if (!can_split) return pl;
    if (max_splits > 1) {
        const int bm_s = bm_env == 256 && M > 128 ? 256 : 128;
        const int64_t nt_f = ((M + bm_s - 1) / bm_s) * tiles_n;
        const int sp = (int)std::min<int64_t>(std::min<int64_t>(max_splits, P), CU / std::max<int64_t>(nt_f, 1));
        if (sp >= 2) { pl.bm = bm_s; pl.splits = sp; }
        return pl;
    }
    if (P < 8) return pl;
    double best = pl.t_us / 1.05;
    for (int bm_c = 128; bm_c <= 256; bm_c += 128) {
        if (bm_c == 256 && (M <= 128 || bm_env == 128)) continue;
        if (bm_c == 128 && bm_env == 256) continue;
        const int64_t nt_c = bm_c == 256 ? nt256 : nt128;
        if (nt_c > CU / 2) continue;
        for (int s_ = 2; s_ <= 32 && s_ * nt_c <= CU && 2 * s_ <= P; ++s_) {
            const double t = 12.5 + mm_stage_us(bm_c, (double)(s_ * nt_c), CU) * (double)((P + s_ - 1) / s_) + 0.014 * (double)(s_ * nt_c * (bm_c / 128));
            if (t < best) { best = t; pl.bm = bm_c; pl.splits = s_; pl.t_us = t; }
        }
    }
    return pl;
}

// What the plan of a launch of M x N outputs over P stages predicts (microseconds): the convolution front end chooses between its two row enumerations with it.
// ex: the launch's extras, or nullptr (no split-K scratch).
double gsw_mm_predict_us(int64_t M, int N, int P, const GswMmExtras* ex) {
    const bool have_ws = ex && ex->workspace_bytes > 0 && ex->workspace_dev;
    const int max_splits = ex ? ex->max_splits : 0;
    const MMPlan pl = mm_plan(M, ((int64_t)N + 159) / 160, P, have_ws && max_splits != 1, max_splits);
    const int64_t need = (int64_t)pl.splits * (((M + pl.bm - 1) / pl.bm) * (((int64_t)N + 159) / 160)) * 8 * 5 * (pl.bm / 64) * 64 * 16;
    if (pl.splits >= 2 && need > (ex ? ex->workspace_bytes : 0)) return mm_plan(M, ((int64_t)N + 159) / 160, P, false, 1).t_us;
    return pl.t_us;
}

// Launch the engine for a prepared MMArgs (segments, weights, epilogue); fills the tiling fields.
int gsw_mm_launch(MMArgs& a, int dtype, void* stream, GswMmExtras* ex) {
    GswMmExtras none;
    if (!ex) { gsw_mm_no_extras(&none); ex = &none; }
    else { ex->colstats_rows_per_block = 0; ex->colstats_blocks = 0; ex->rowstats_slots = 0; ex->splits = 1; }
    if (dtype != GSW_F16 && dtype != GSW_BF16) return GSW_ERR_BAD_ARG;
    if (ex->colstats_capacity < 0 || ex->rowstats_capacity < 0 || ex->workspace_bytes < 0 || ex->max_splits < 0 || ex->max_splits > 64
        || ((uintptr_t)ex->colstats_dev & 15) || ((uintptr_t)ex->rowstats_dev & 7) || ((uintptr_t)ex->workspace_dev & 15)
        || (ex->flags & ~GSW_MM_GN_ONLY)) return GSW_ERR_BAD_ARG;      // (unknown flag bits: a caller that filled the struct field by field without zeroing it)
    void* const ws_dev = ex->workspace_bytes > 0 ? ex->workspace_dev : nullptr;
    const int64_t ws_bytes = ws_dev ? ex->workspace_bytes : 0;
    const int max_splits = ex->max_splits;
    // N: any multiple of 8 (the last 160-column tile may be partial: weight rows are clamped, stores masked); GEGLU pairs columns inside a tile
    if (a.N % 8 || (a.mode == MM_MODE_GEGLU && a.N % 160) || a.M <= 0 || a.P <= 0) return GSW_ERR_UNSUPPORTED;
    if (a.mode == MM_MODE_QKV && (a.n_rows <= 0 || a.n_rows % 160 || a.n_rows >= a.N || !a.y2)) return GSW_ERR_UNSUPPORTED;
    constexpr int BN = 160;
    const int64_t tiles_n = (a.N + BN - 1) / BN;
    // panel of the tile order: 8 column tiles, or all of them when their weight tiles (tiles_n x 160 rows x K) stay under ~2 MiB of an XCD's 4 MiB L2
    {
        static const int panel_env = getenv("GSW_MM_PANEL") ? atoi(getenv("GSW_MM_PANEL")) : 0;       // A/B switch: 8 = the fixed panel of ABI < 0.4.0
        a.panel = 8;
        if (tiles_n > 8 && tiles_n <= 32 && tiles_n * BN * (int64_t)a.P * 64 * 2 <= (2 << 20)) a.panel = (int32_t)tiles_n;
        if (panel_env > 0) a.panel = panel_env;
    }
    const MMPlan plan = mm_plan(a.M, tiles_n, a.P, ws_dev && max_splits != 1 && !a.ln_stat, max_splits);       // tiling and split-K policy: see mm_plan
    const int BM = plan.splits >= 2 ? mm_plan(a.M, tiles_n, a.P, false, 1).bm : plan.bm;      // the tile of the UNSPLIT launch (also taken when a split plan does not fit the workspace)
    hipStream_t st = (hipStream_t)stream;
    a.splits = 1; a.ws = nullptr;
    // the dense-row / GEGLU epilogues fetch the bias by 16-byte LDS-DMA pieces (STG in the kernel)
    if ((a.mode == MM_MODE_DENSE || a.mode == MM_MODE_GEGLU) && ((uintptr_t)a.bias & 15u)) return GSW_ERR_BAD_ARG;
    float* const cs_req = ex->colstats_capacity > 0 ? ex->colstats_dev : nullptr;
    const int64_t cs_cap = cs_req ? ex->colstats_capacity : 0;
    float* const rs_req = ex->rowstats_capacity > 0 ? ex->rowstats_dev : nullptr;
    const int64_t rs_cap = rs_req ? ex->rowstats_capacity : 0;
    a.colstats = nullptr;
    a.rowstats = nullptr;
    if (a.ln_stat && (a.mode != MM_MODE_DENSE && a.mode != MM_MODE_GEGLU && a.mode != MM_MODE_TRANS)) return GSW_ERR_UNSUPPORTED;
    if (a.ln_stat && a.rowbias) return GSW_ERR_UNSUPPORTED;
    // Split-K for launches that cannot fill the chip with output tiles (the deep levels at small batch: 8 x 8 pixels of one image are ONE row tile
    // against 180-360 K stages): `splits` workgroups share a tile's stages, fp32 partials go through the caller's workspace, a second small kernel
    // adds them in a fixed order and runs the epilogue.  Needs a workspace (gsw_mm_set_workspace); without one the launch runs unsplit.
    if (plan.splits >= 2) {
        const int bm_s = plan.bm, splits = plan.splits;
        const int64_t nt = (((int64_t)a.M + bm_s - 1) / bm_s) * tiles_n;
        const int64_t need = (int64_t)splits * nt * 8 * 5 * (bm_s / 64) * 64 * 16;
        if (need <= ws_bytes) {
            a.tiles_n = (int32_t)tiles_n;
            a.ntiles = (int32_t)nt;
            a.splits = splits; a.ws = (float*)ws_dev;
            ex->splits = splits;
            const uint32_t grid = (uint32_t)((nt * splits + 7) / 8 * 8);
            const int e = dtype == GSW_F16 ? mm_launch_splitk<_Float16>(a, grid, bm_s / 64, st) : mm_launch_splitk<__bf16>(a, grid, bm_s / 64, st);
            if (e != 0) { g_last_hip_error = e; return GSW_ERR_HIP; }
            return GSW_OK;
        }
    }
    // The wide tile (256 x 320, MT = 8): launches with enough of those tiles to keep every CU busy for several rounds and a K loop long enough to amortise
    // the longer fill (two 72 KiB stages).  Fewer operand bytes per MFMA is what pays under the board's power limit.
    // GSW_MM_WIDE=0 / gsw_mm_config(tile_rows = 256 or 128) keep the narrower tiles (A/B, tests); tile_rows = 512 forces the wide tile wherever it is legal.
    bool wide = false;
    {
        static const int wide_mask = getenv("GSW_MM_WIDE") ? atoi(getenv("GSW_MM_WIDE")) : 7;       // bit e: epilogue kind e (0 dense rows, 1 PF rows, 2 GEGLU) may take the wide tile; 0 = never (A/B)
        const int epi_k = a.mode == MM_MODE_GEGLU ? 2 : (a.mode == MM_MODE_DENSE && !a.rowbias) ? 0 : 1;
        const int wide_env = (wide_mask >> epi_k) & 1;
        const int bm_cfg = g_mm_tile_rows.load(std::memory_order_relaxed);
        const bool mode_ok = a.mode == MM_MODE_DENSE || a.mode == MM_MODE_PF || a.mode == MM_MODE_TOK2PF || a.mode == MM_MODE_UP2X || a.mode == MM_MODE_GEGLU;
        const int64_t tn_w = (a.N + 319) / 320, tm_w = ((int64_t)a.M + 255) / 256;
        // buffer addressing of the wide producer: no weight-row clamp (N % 320 == 0), every activation segment below 4 GiB
        int64_t rows_in = a.M;
        if (a.mode == MM_MODE_PF || a.mode == MM_MODE_UP2X) {
            const int64_t per_img = (a.flags & MM_FLAG_COMPACT) ? (int64_t)std::max(1, (a.Hp - 2) * (a.Wp - 2)) : (int64_t)a.Hp * a.Wp;
            rows_in = ((int64_t)a.M / per_img + 1) * (int64_t)a.in_Hp * a.in_Wp + 2 * (int64_t)a.in_Wp + 4;
        }
        int64_t ld_max = 0;
        for (int i = 0; i < a.nseg; ++i) ld_max = std::max<int64_t>(ld_max, a.seg[i].ld);
        // rows of the output / residual row space (the residual touches address it through a buffer descriptor too)
        int64_t rows_out = a.M;
        if (a.mode == MM_MODE_PF || a.mode == MM_MODE_TOK2PF) rows_out = ((int64_t)a.M / std::max<int64_t>(1, (a.mode == MM_MODE_TOK2PF ? a.S : ((a.flags & MM_FLAG_COMPACT) ? (int64_t)(a.Hp - 2) * (a.Wp - 2) : (int64_t)a.Hp * a.Wp))) + 1) * (int64_t)a.Hp * a.Wp;
        if (a.mode == MM_MODE_UP2X) rows_out = rows_in * 4 + 8;
        const bool res_ok = !a.resid || rows_out * (int64_t)a.ldr * 2 < ((int64_t)1 << 32) - (1 << 20);
        // (the dense-row epilogue of the wide tile addresses its OUTPUT by a 32-bit byte offset too)
        const bool y_ok = !(a.mode == MM_MODE_DENSE && !a.rowbias) || (int64_t)a.M * a.ldy * 2 < ((int64_t)1 << 32) - (1 << 20);
        const bool legal = mode_ok && res_ok && y_ok && a.N >= 320 && a.N % 320 == 0 && (!(a.mode == MM_MODE_GEGLU || (a.mode == MM_MODE_DENSE && !a.rowbias)) || a.M % 256 == 0) && rows_in * ld_max * 2 < ((int64_t)1 << 32) - (1 << 20) && (int64_t)a.N * a.ldw * 2 < ((int64_t)1 << 32) - (1 << 20);
        // a partial last column tile costs a whole one: at most 1/8 of the column tiles' work wasted
        // measured per shape at 128 rows (profiles/r05g_unet_forward_b128_wide_thresholds.txt): the dense-row and GEGLU launches win at every K of the eps model,
        // K = 320 included (-5 ... -24 %: a 320-column tile reads the activations once where two 160-column tiles read them twice) -- except the K = 320 launches
        // WITH a residual operand (+6 %: five stages do not pay for the longer epilogue), which stay narrow; the PF-row epilogue (convolutions, token scatter:
        // per-row residual / row-bias fetches, twice as long per wave on the wide tile) needs a longer K loop -- 3 x 3 convolutions win from K = 5760 on
        // (-3 ... -7 %) and lose 2-6 % at K = 2880
        static const int pmin_dense = getenv("GSW_MM_WIDE_PMIN") ? atoi(getenv("GSW_MM_WIDE_PMIN")) : 5;          // (A/B knobs: stages from which the dense-row / GEGLU
        static const int pmin_pf = getenv("GSW_MM_WIDE_PMIN_PF") ? atoi(getenv("GSW_MM_WIDE_PMIN_PF")) : 64;      //  and the PF-row launches take the wide tile)
        static const int pmin_res = getenv("GSW_MM_WIDE_PMIN_RES") ? atoi(getenv("GSW_MM_WIDE_PMIN_RES")) : 8;     //  (dense rows with a residual operand)
        const int p_min = epi_k == 1 ? pmin_pf : (a.resid ? std::max(pmin_dense, pmin_res) : pmin_dense);
        // rounds of 256 workgroups: a stage of a wide tile costs 1.77 x a stage of a 256 x 160 tile for 2 x its outputs (1.70 vs 0.96 us, the slopes of time against K on
        // the 64 x 64 convolutions) -- wide wins when its rounds, at that price, are fewer than the narrow tiling's (a half-empty last round can eat the gain:
        // 4.5 rounds of wide tiles against 9 of narrow ones still win, 2.25 against 4.5 do not)
        const int64_t cus = mm_cus();
        const int64_t t_w = tm_w * tn_w, t_n = (((int64_t)a.M + 255) / 256) * tiles_n;
        const double cost_w = 1.77 * (double)((t_w + cus - 1) / cus), cost_n = (double)((t_n + cus - 1) / cus);
        const bool fits = t_w >= cus && cost_w <= 0.995 * cost_n && a.P >= p_min && a.M >= 2048;
        wide = legal && (bm_cfg == 512 || (bm_cfg == 0 && wide_env != 0 && fits));
    }
    const int BMt = wide ? 256 : BM, BNt = wide ? 320 : BN;
    const int64_t tiles_nt = (a.N + BNt - 1) / BNt;
    const int64_t tiles_m = ((int64_t)a.M + BMt - 1) / BMt;
    if (tiles_m * tiles_nt > 0x7FFFFFFF) return GSW_ERR_UNSUPPORTED;
    a.tiles_n = (int32_t)tiles_nt;
    a.ntiles = (int32_t)(tiles_m * tiles_nt);
    if (wide) {          // panel of the tile order for 320-column tiles: 4 (the same 1280 columns), or all of them under the same L2 budget
        a.panel = 4;
        if (tiles_nt > 4 && tiles_nt <= 16 && tiles_nt * 320 * (int64_t)a.P * 64 * 2 <= (2 << 20)) a.panel = (int32_t)tiles_nt;
    }
    const uint32_t grid = (uint32_t)std::min<int64_t>(mm_cus(), (a.ntiles + 7) / 8 * 8);
    const int ngrp = wide ? 4 : 2, wmv = wide ? 2 : 4;        // 80-column groups per tile, waves along M
    // row statistics: plain dense-row launches (EPI 0), unsplit
    if (rs_req && a.mode == MM_MODE_DENSE && !a.rowbias && !a.ln_stat && (int64_t)a.M * ngrp * tiles_nt * 2 <= rs_cap) {
        a.rowstats = rs_req;
        ex->rowstats_slots = (int)(ngrp * tiles_nt);
    }
    // column statistics: EPI 1 launches whose M dimension enumerates real pixels / tokens (interior enumeration or the token scatter), unsplit
    if (cs_req && (a.mode == MM_MODE_TOK2PF || ((a.mode == MM_MODE_PF || a.mode == MM_MODE_UP2X) && (a.flags & MM_FLAG_COMPACT)))
        && tiles_m * wmv * (int64_t)a.N <= cs_cap) {
        a.colstats = cs_req;
        ex->colstats_rows_per_block = BMt / wmv; ex->colstats_blocks = (int)(tiles_m * wmv);
    }
    const int epi = a.mode == MM_MODE_QKV ? 5 : a.mode == MM_MODE_TRANS ? 3 : a.mode == MM_MODE_GEGLU ? 2 : (a.mode == MM_MODE_DENSE && !a.rowbias) ? 0 : 1;
    const int e = dtype == GSW_F16 ? mm_launch_e<_Float16>(a, epi, grid, wide ? 8 : BM / 64, st) : mm_launch_e<__bf16>(a, epi, grid, wide ? 8 : BM / 64, st);
    if (e != 0) { g_last_hip_error = e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_gemm_strided(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr,
                     void* y_dev, int64_t ldy, int64_t M, int K, int N, int mode, int S, int Wimg, int dtype, void* stream) {
    return gsw_gemm_ex(x_dev, ldx, w_dev, ldw, bias_dev, resid_dev, ldr, y_dev, ldy, M, K, N, mode, S, Wimg, dtype, nullptr, stream);
}

int gsw_gemm_ex(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr, void* y_dev, int64_t ldy,
                int64_t M, int K, int N, int mode, int S, int Wimg, int dtype, GswMmExtras* ex, void* stream) {
    if (!x_dev || !w_dev || !y_dev || M <= 0 || K <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if (mode != GSW_GEMM_PLAIN && mode != GSW_GEMM_GEGLU && mode != GSW_GEMM_TRANS && mode != GSW_GEMM_TOK2PF) return GSW_ERR_BAD_ARG;
    if (K % 64 || N % 8 || (mode == GSW_GEMM_GEGLU && N % 160) || M > 0x7FFFFF00 || ldx < K || ldw < K || (ldx & 7) || (ldw & 7) || (ldy & 7) || (ldr & 7)
        || M * ldx >= ((int64_t)1 << 40) || (int64_t)N * ldw >= ((int64_t)1 << 31) || ldx >= ((int64_t)1 << 31) || ldy >= ((int64_t)1 << 31) || ldr >= ((int64_t)1 << 31))
        return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_GEGLU && resid_dev) return GSW_ERR_UNSUPPORTED;
    if ((mode == GSW_GEMM_PLAIN || mode == GSW_GEMM_GEGLU) && ((uintptr_t)bias_dev & 15u)) return GSW_ERR_BAD_ARG;       // fetched by 16-byte LDS-DMA pieces
    if (mode == GSW_GEMM_TRANS && (resid_dev || S <= 0 || (S & 7) || M % S)) return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_TOK2PF && (S <= 0 || Wimg <= 0 || S % Wimg || M % S)) return GSW_ERR_BAD_ARG;
    const int64_t ncols = mode == GSW_GEMM_GEGLU ? N / 2 : N;
    if (mode != GSW_GEMM_TRANS && (ldy < ncols || (resid_dev && ldr < ncols))) return GSW_ERR_BAD_ARG;
    MMArgs a;
    for (int i = 0; i < 3; ++i) a.seg[i] = MMSeg{x_dev, (int32_t)ldx, K / 64, 1, 1, 0, 0, 0};
    a.nseg = 1; a.P = K / 64;
    a.w = w_dev; a.ldw = (int32_t)ldw;
    a.M = (int32_t)M; a.N = N;
    a.bias = bias_dev; a.rowbias = nullptr; a.resid = resid_dev; a.y = y_dev; a.colstats = nullptr; a.y2 = nullptr; a.n_rows = 0;
    a.ln_stat = nullptr; a.ln_u = nullptr; a.ln_v = nullptr;
    a.ldy = (int32_t)ldy; a.ldr = (int32_t)ldr; a.ldrb = N;
    a.Hp = 1; a.Wp = 1; a.in_Hp = 1; a.in_Wp = 1; a.stride = 1; a.S = S > 0 ? S : 1; a.Wimg = Wimg > 0 ? Wimg : 1; a.up = 0; a.flags = MM_FLAG_NONE;
    a.mode = MM_MODE_DENSE;
    if (mode == GSW_GEMM_GEGLU) a.mode = MM_MODE_GEGLU;
    else if (mode == GSW_GEMM_TRANS) a.mode = MM_MODE_TRANS;
    else if (mode == GSW_GEMM_TOK2PF) { a.mode = MM_MODE_TOK2PF; a.Wp = Wimg + 2; a.Hp = S / Wimg + 2; }
    return gsw_mm_launch(a, dtype, stream, ex);
}

int gsw_gemm_qkv(const void* x_dev, const void* w_dev, const void* bias_dev, void* rows_dev, void* trans_dev, int64_t M, int K, int N_rows, int N,
                 int S, int dtype, void* stream) {
    // one pass over x [M, K]: columns [0, N_rows) of x w^T (+ bias) -> rows_dev [M, N_rows] (self-attention's q | k), columns [N_rows, N) -> trans_dev
    // [M / S][N - N_rows][S] (the value projection transposed, what gsw_attention consumes).  w [N][K]; N_rows % 160 == 0; S % 8 == 0; M % S == 0.
    if (!x_dev || !w_dev || !rows_dev || !trans_dev || M <= 0 || K <= 0 || N <= 0 || N_rows <= 0 || N_rows >= N || S <= 0) return GSW_ERR_BAD_ARG;
    if (K % 64 || N % 8 || N_rows % 160 || (S & 7) || M % S || M > 0x7FFFFF00 || M * (int64_t)K >= ((int64_t)1 << 40) || (int64_t)N * K >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    MMArgs a;
    for (int i = 0; i < 3; ++i) a.seg[i] = MMSeg{x_dev, K, K / 64, 1, 1, 0, 0, 0};
    a.nseg = 1; a.P = K / 64;
    a.w = w_dev; a.ldw = K;
    a.M = (int32_t)M; a.N = N;
    a.bias = bias_dev; a.rowbias = nullptr; a.resid = nullptr; a.y = rows_dev; a.colstats = nullptr; a.y2 = trans_dev; a.n_rows = N_rows;
    a.ln_stat = nullptr; a.ln_u = nullptr; a.ln_v = nullptr;
    a.ldy = N_rows; a.ldr = N_rows; a.ldrb = N;
    a.Hp = 1; a.Wp = 1; a.in_Hp = 1; a.in_Wp = 1; a.stride = 1; a.S = S; a.Wimg = 1; a.up = 0; a.flags = MM_FLAG_NONE;
    a.mode = MM_MODE_QKV;
    return gsw_mm_launch(a, dtype, stream, nullptr);
}

// Row records [M][slots][2] (sum, sum of squares per 80-column half tile) -> (rstd, -rstd * mean) per row, the form the LayerNorm-folded epilogues read
__global__ __launch_bounds__(256) void gsw_ln_rowstats_finish_kernel(const float2* __restrict__ rec, int32_t slots, int64_t M, float inv_c, float eps, float2* __restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f, q = 0.f;
    for (int32_t k = 0; k < slots; ++k) { const float2 v = rec[m * slots + k]; s += v.x; q += v.y; }
    const float mean = s * inv_c;
    const float var = fmaxf(q * inv_c - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    out[m] = make_float2(rstd, -rstd * mean);
}

int gsw_ln_rowstats_finish(const float* records_dev, int slots, int64_t M, int C, float eps, float* stat_dev, void* stream) {
    if (!records_dev || !stat_dev || slots <= 0 || M <= 0 || C <= 0) return GSW_ERR_BAD_ARG;
    hipLaunchKernelGGL(gsw_ln_rowstats_finish_kernel, dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(records_dev), slots, M,
                       1.0f / (float)C, eps, reinterpret_cast<float2*>(stat_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return GSW_ERR_HIP; }
    return GSW_OK;
}

int gsw_gemm_ln(const void* x_dev, const float* ln_stat_dev, const void* w_dev, const float* u_dev, const float* v_dev, void* y_dev, int64_t M, int K, int N,
                int mode, int S, int dtype, void* stream) {
    return gsw_gemm_ln_ex(x_dev, ln_stat_dev, w_dev, u_dev, v_dev, y_dev, M, K, N, mode, S, dtype, nullptr, stream);
}

int gsw_gemm_ln_ex(const void* x_dev, const float* ln_stat_dev, const void* w_dev, const float* u_dev, const float* v_dev, void* y_dev, int64_t M, int K, int N,
                   int mode, int S, int dtype, GswMmExtras* ex, void* stream) {
    // y = LayerNorm(x) W^T + b without materialising LayerNorm(x): w_dev = W diag(gamma) [N][K], u = (row sums of w_dev), v = W beta + b (fp32 [N]),
    // ln_stat_dev float2 [M] = (rstd, -rstd * mean) of the rows of x (gsw_ln_rowstats_finish).  mode: GSW_GEMM_PLAIN / GEGLU (packed rows) / TRANS.
    if (!x_dev || !ln_stat_dev || !w_dev || !u_dev || !v_dev || !y_dev || M <= 0 || K <= 0 || N <= 0) return GSW_ERR_BAD_ARG;
    if (mode != GSW_GEMM_PLAIN && mode != GSW_GEMM_GEGLU && mode != GSW_GEMM_TRANS) return GSW_ERR_BAD_ARG;
    if (K % 64 || N % 8 || (mode == GSW_GEMM_GEGLU && N % 160) || (M & 7) || M > 0x7FFFFF00 || M * (int64_t)K >= ((int64_t)1 << 40) || (int64_t)N * K >= ((int64_t)1 << 31)) return GSW_ERR_UNSUPPORTED;
    if (mode == GSW_GEMM_TRANS && (S <= 0 || (S & 7) || M % S)) return GSW_ERR_UNSUPPORTED;
    if (((uintptr_t)u_dev | (uintptr_t)v_dev | (uintptr_t)ln_stat_dev) & 15) return GSW_ERR_BAD_ARG;
    MMArgs a;
    for (int i = 0; i < 3; ++i) a.seg[i] = MMSeg{x_dev, K, K / 64, 1, 1, 0, 0, 0};
    a.nseg = 1; a.P = K / 64;
    a.w = w_dev; a.ldw = K;
    a.M = (int32_t)M; a.N = N;
    a.bias = nullptr; a.rowbias = nullptr; a.resid = nullptr; a.y = y_dev; a.colstats = nullptr; a.y2 = nullptr; a.n_rows = 0;
    a.ln_stat = ln_stat_dev; a.ln_u = u_dev; a.ln_v = v_dev;
    const int ncols = mode == GSW_GEMM_GEGLU ? N / 2 : N;
    a.ldy = ncols; a.ldr = ncols; a.ldrb = N;
    a.Hp = 1; a.Wp = 1; a.in_Hp = 1; a.in_Wp = 1; a.stride = 1; a.S = S > 0 ? S : 1; a.Wimg = 1; a.up = 0; a.flags = MM_FLAG_NONE;
    a.mode = mode == GSW_GEMM_GEGLU ? MM_MODE_GEGLU : mode == GSW_GEMM_TRANS ? MM_MODE_TRANS : MM_MODE_DENSE;
    return gsw_mm_launch(a, dtype, stream, ex);
}

int gsw_gemm(const void* x_dev, const void* w_dev, const void* bias_dev, const void* resid_dev, void* y_dev, int64_t M, int K, int N,
             int mode, int S, int Wimg, int dtype, void* stream) {
    return gsw_gemm_strided(x_dev, K, w_dev, K, bias_dev, resid_dev, N, y_dev, mode == GSW_GEMM_GEGLU ? N / 2 : N, M, K, N, mode, S, Wimg, dtype, stream);
}
