/* gswm.h -- C ABI of the MI355X-native Gaussian-Shading watermark hot path (libgswm.so).
 *
 * Every entry point is what a binding of the reference's hot path would call; the reference
 * (lthero-big/A-watermark-for-Diffusion-Models @ 2024_08_07) is pure Python, so "the interface this
 * replaces" is a Python function body, cited as file:line relative to the reference root.
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes; no torch / C++ types cross the boundary
 *   - `*_dev` pointers are DEVICE (HBM) pointers owned by the caller; all other pointers are HOST
 *     pointers that are only read during the call (key, nonce, message are copied into the kernel
 *     arguments)
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is stream-ordered
 *     and asynchronous, never synchronises, never allocates device memory (exception: gsw_embed with a
 *     message longer than GSW_MSG_INLINE_MAX bytes stages it with hipMallocAsync/hipFreeAsync)
 *   - return value: 0 = GSW_OK, otherwise a gsw_status; no exceptions cross the boundary; no global
 *     mutable state (the in-kernel RNG is fully specified by seed + image index)
 *   - thread-safe for concurrent calls on distinct streams
 *   - per-image lattice is the C-contiguous [4, H/8, W/8] latent, addressed by flat element index
 *     i in [0, n_elems); batches are [B, n_elems] contiguous
 */
#ifndef GSWM_H
#define GSWM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libgswm.so is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GSW_VERSION 500 /* 0.5.0: every engine launch takes what it needs besides its operands in a caller-owned GswMmExtras (gsw_*_ex; the plain entry points request nothing); the thread-local one-shot side channels of ABI 0.3 (gsw_mm_next_colstats / gsw_mm_last_colstats / gsw_mm_next_rowstats / gsw_mm_last_rowstats / gsw_mm_set_workspace) and the round-1 aliases gsw_linear / gsw_attention_hd64 are gone; gsw_xattn_fused is new */

#define GSW_MSG_INLINE_MAX 256 /* message bytes carried inside the kernel arguments (2048 bit) */

typedef enum gsw_dtype {
    GSW_F32 = 0,
    GSW_F16 = 1,
    GSW_BF16 = 2,
    GSW_F64 = 3
} gsw_dtype;

typedef enum gsw_status {
    GSW_OK = 0,
    GSW_ERR_BAD_ARG = 1,     /* null pointer, bad dtype, non-positive size, n_elems % 4 != 0 ...        */
    GSW_ERR_UNSUPPORTED = 2, /* lattice too large for the LDS-resident vote             */
    GSW_ERR_RAGGED = 3,      /* 8*ceil(n_elems/8) is not a multiple of msg_bits: the reference raises
                                IndexError at extract.py:98                                             */
    GSW_ERR_HIP = 4,         /* a HIP runtime call failed; gsw_last_hip_error() has the hipError_t      */
    GSW_WARN_NO_RECORDS = 5  /* (ABI < 0.5.0: the one-shot record queries reported a dropped request with it; a launch now says what it wrote in
                                GswMmExtras.colstats_rows_per_block / rowstats_slots -- 0 means: take the statistics pass instead) */
} gsw_status;

/* embed flags */
#define GSW_EMBED_EXACT_F64 0u /* Cephes ndtri evaluated in fp64, then rounded (== scipy to <= a few fp64 ulp) */
#define GSW_EMBED_FAST_F32 1u  /* fp32 core on the tail-safe side of the distribution; |dz| <= 1e-5       */

/* per-image flag bits written by gsw_extract */
#define GSW_FLAG_SATURATED 1u /* some z >= 8.292361075813597: norm.cdf == 1.0, y == 2 (extract.py:84-86 raises ValueError) */
#define GSW_FLAG_NAN 2u       /* some z is NaN (int(nan) raises ValueError at extract.py:84)            */

int gsw_version(void);
/* Measurement switches the library was COMPILED with (csrc/gswm_ablate.inc: cycle stamps, ablations of the matmul engine's main loop -- builds that compute
 * wrong results by design).  A shippable library returns 0; tests/test_kernel_metadata.py and the loader (_native.lib) refuse anything else. */
int gsw_build_flags(void);
const char* gsw_strerror(int status);
int gsw_last_hip_error(void); /* thread-local hipError_t of the last GSW_ERR_HIP on this thread */

/* E2 -- gs_insert.py:45-47 / extract.py:77-78,87: the ChaCha20 stream `cryptography` (OpenSSL) produces for
 * Cipher(algorithms.ChaCha20(key, nonce16)): nonce16[0:4] = LE32 initial block counter (carry into the next
 * word), nonce16[4:16] = RFC 8439 nonce.  Writes `nbytes` keystream bytes to out_dev. */
int gsw_keystream(const uint8_t key[32], const uint8_t nonce16[16], uint8_t* out_dev, size_t nbytes, void* stream);

/* E1-E6 -- gs_insert.py:8-66 (and the generalised lattice of ComfyUI_GSWaterMark/nodes.py:51-123), batched.
 *   msg/msg_bytes : the padded watermark k (gs_insert.py:9-20); plaintext = k repeated floor(n_elems/(8*msg_bytes))
 *                   times then zeros (nodes.py:79-87)
 *   u_dev         : optional [B, n_elems] float64 uniforms in [0,1) (the reference's np.random.uniform draws,
 *                   gs_insert.py:62) -- bit-parity mode.  NULL => in-kernel Philox4x32-7 (Random123 philox4x32_R(7)) keyed by `seed`,
 *                   counter = (element index >> 2, image_index0 + b), one 32-bit word per element, u = (w + 1/2) 2^-32:
 *                   results do not depend on batch split
 *                   or GPU count.
 *   out_dev       : [B, n_elems] of out_dtype; z = ndtri((u + y) / 2) (gs_insert.py:64), y = cipher bit,
 *                   MSB-first within each cipher byte (gs_insert.py:49)
 *   n_elems       : 4 * (H/8) * (W/8); must be a multiple of 4 */
int gsw_embed(const uint8_t key[32], const uint8_t nonce16[16], const uint8_t* msg, int msg_bytes,
              const double* u_dev, uint64_t seed, uint64_t image_index0, void* out_dev, int out_dtype, int B,
              int64_t n_elems, uint32_t flags, void* stream);

/* In-kernel RNG only: writes the u the kernel would draw ([B, n_elems] float64) so a CPU oracle can be run on
 * exactly the same inputs. */
int gsw_philox_uniform(uint64_t seed, uint64_t image_index0, double* u_dev, int B, int64_t n_elems, void* stream);

/* X3-X5 -- extract.py:72-101, batched: y = int(norm.cdf(float64(z)) * 2) per element in C order, pack MSB-first,
 * ChaCha20-decrypt, split into msg_bits-wide segments, strict-majority vote per bit (ties -> 0).
 *   z_dev      : [B, n_elems] of z_dtype
 *   bits_dev   : [B, ceil(msg_bits/8)] recovered message, MSB-first (bit t -> byte t>>3, bit 7-(t&7))
 *   counts_dev : optional [B, msg_bits] uint32 number of '1' votes per bit (NULL to skip)
 *   flags_dev  : [B] uint32 GSW_FLAG_* (the reference raises for such images; the host wrapper does too)
 * Returns GSW_ERR_RAGGED when the reference would raise IndexError. */
int gsw_extract(const void* z_dev, int z_dtype, const uint8_t key[32], const uint8_t nonce16[16], int msg_bits,
                uint8_t* bits_dev, uint32_t* counts_dev, uint32_t* flags_dev, int B, int64_t n_elems, void* stream);

/* X6 -- extract.py:103-110 on device: matches_dev[b] = number of equal bits between bits_dev[b] and the
 * reference message over the first min(msg_bits, ref_bits) positions. */
int gsw_bit_matches(const uint8_t* bits_dev, int msg_bits, const uint8_t* ref_msg, int ref_bits,
                    uint32_t* matches_dev, int B, void* stream);

/* X2 / G1 -- the elementwise update of the DDIM (eta = 0) sampling / inversion loop
 * (inverse_stable_diffusion_gs.pyc `backward_ddim`; diffusers DDIMScheduler/DDIMInverseScheduler.step):
 *   out = a * x + b * model_out            (a, b precomputed per step on the host in fp64)
 * computed in fp32, one rounding to `dtype`.  out_dev may alias x_dev. */
int gsw_ddim_step(const void* x_dev, const void* model_out_dev, void* out_dev, float a, float b, int dtype,
                  int64_t n, void* stream);

/* G1 with classifier-free guidance fused (modified_stable_diffusion_gs.pyc: uncond + g * (text - uncond), then the
 * scheduler step): out = a * x + b * (e_uncond + g * (e_text - e_uncond)). */
int gsw_ddim_step_cfg(const void* x_dev, const void* e_uncond_dev, const void* e_text_dev, void* out_dev, float a,
                      float b, float guidance, int dtype, int64_t n, void* stream);

/* Last inversion step fused with the extract tail: z = a * x + b * model_out is voted on directly (and stored to
 * z_out_dev when non-NULL, in `dtype`, after the same rounding an unfused gsw_ddim_step would apply). */
int gsw_ddim_step_extract(const void* x_dev, const void* model_out_dev, void* z_out_dev, float a, float b, int dtype,
                          const uint8_t key[32], const uint8_t nonce16[16], int msg_bits, uint8_t* bits_dev,
                          uint32_t* counts_dev, uint32_t* flags_dev, int B, int64_t n_elems, void* stream);

/* X2 / G1 -- elementwise fusions inside the eps model (the UNet the DDIM loops call; in the reference this is diffusers'
 * ResnetBlock2D / GEGLU running as separate torch kernels):
 *   gsw_groupnorm_silu: out = act(GroupNorm_groups(x + pre_bias[b, c]) * gamma[c] + beta[c]); x, out: [B, C, HW] (NCHW), HW % 8 == 0;
 *                       pre_bias: optional [B, C] (the time-embedding projection added before norm2); act: 0 none, 1 SiLU
 *   gsw_geglu:          out[r, i] = in[r, i] * gelu(in[r, inner + i]); in: [rows, 2*inner], out: [rows, inner], inner % 8 == 0 */
int gsw_groupnorm_silu(const void* x_dev, const void* pre_bias_dev, const void* gamma_dev, const void* beta_dev, void* out_dev, int B, int C,
                       int HW, int groups, float eps, int act, int dtype, void* stream);
int gsw_geglu(const void* in_dev, void* out_dev, int64_t rows, int inner, int dtype, void* stream);

/* X2 / G1 -- convolution of the eps model as an MFMA implicit GEMM on "padded-flat NHWC" (PF) activations
 * (replaces diffusers' Conv2d -> MIOpen in ResnetBlock2D / Downsample2D / conv_shortcut):
 *   PF tensor: [B*(H+2)*(W+2)] rows x C channels, row (b, y, x) = b*(H+2)*(W+2) + y*(W+2) + x, zero border, plus (W+3) zero guard
 *              rows before row 0 and after the last row (x_dev points at row 0).
 *   y[m, n] = sum_taps sum_c x[row(m) + off_tap, c] * w[n, tap*C + c] + bias[n] + rowbias[b(m), n] + resid[m, n]; border rows = 0
 *   w_dev: [N][ksize*ksize*C] (tap-major, channel-minor), ksize 1 or 3, stride 1 or 2 (3x3 only); H, W = OUTPUT size;
 *   ldx: row stride of x in elements; C % 64 == 0; N % 8 == 0 from 128 channels up (the matmul engine, csrc/gswm_mm.hip; a partial last 160-column
 *   tile costs a full one), N % 64 == 0 below that (the 64-column kernel of round 1: the 4-channel edges padded to one tile);
 *   dtype GSW_F16 / GSW_BF16; bias / rowbias / resid optional.  rowbias: [B] rows, ld_rowbias elements apart (0 = N; else >= N and a
 *   multiple of 8): the time-embedding projections of ALL resnets come out of one GEMM, each convolution reads its column slice. */
int gsw_conv_pf(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                int B, int H, int W, int C, int N, int ksize, int stride, int ldx, int dtype, void* stream);

/* GroupNorm (+SiLU) on a PF tensor: out = act(GroupNorm(x) * gamma + beta); out is a PF tensor (zero border) or, with out_tokens = 1,
 * dense tokens [B, H*W, C] (the transformer's input).  workspace_dev: >= B * 64 * groups * 2 floats.  C % 8 == 0, groups <= 64. */
int gsw_groupnorm_pf(const void* x_dev, const void* gamma_dev, const void* beta_dev, void* out_dev, float* workspace_dev, int B, int H, int W,
                     int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream);

/* Same with two sources: GroupNorm over the channel concatenation [x (Ca channels) | x2 (C - Ca channels)] read in place (the UNet's
 * skip connections are never materialised as a torch.cat). */
int gsw_groupnorm_pf2(const void* x_dev, const void* x2_dev, int Ca, const void* gamma_dev, const void* beta_dev, void* out_dev, float* workspace_dev,
                      int B, int H, int W, int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream);

/* ---- Extras of a matmul-engine launch (ABI 0.4.0; the only form since 0.5.0).  Everything a launch needs besides its operands travels in ONE caller-owned
 * struct: no thread-local "arm, then launch" state, nothing shared between two streams of one thread, and what the launch actually did comes back in the
 * same struct.  NULL in place of a GswMmExtras* means "no records, no split-K".
 *   colstats_dev / colstats_capacity : in  -- GroupNorm statistics without a pass over the tensor: a convolution / token-scatter launch (gsw_conv_pf_ex,
 *                          gsw_conv3x3_res_pf_ex, gsw_conv_up2x_pf_ex, gsw_gemm_ex with GSW_GEMM_TOK2PF) also writes, per block of 32 or 64 consecutive output pixels and per
 *                          PAIR of output columns (2c, 2c + 1), the sum and the sum of squares of the values it stores: [npar][blocks][2 planes: sums | sums of squares][N / 2]
 *                          floats, npar = 1 (4 for gsw_conv_up2x_pf_ex: one quarter of the buffer per parity launch, records over the low-resolution pixels);
 *                          capacity >= npar * ceil(M / 128) * 4 * N floats covers either tile height (M = output pixels of ONE launch); 16-byte aligned (NULL: none)
 *   rowstats_dev / rowstats_capacity : in  -- LayerNorm statistics likewise: a plain gsw_gemm_ex (+ residual) also writes, per output row and 80-column half tile, the
 *                          (sum, sum of squares) of what it stores: [M][slots][2] floats, capacity >= M * 2 ceil(N / 160) * 2 (NULL: none)
 *   workspace_dev / workspace_bytes / max_splits : in -- split-K scratch (caller-owned device memory, 16-byte aligned, on the device of the launch's stream) and policy.
 *                          Small-batch launches of the eps model -- one image's 8 x 8 level is a single 64-row tile against 180-360 K stages (the reference's own use:
 *                          extract.py:112-117 inverts ONE latent per call) -- cannot fill 256 CUs with output tiles.  With a workspace, a launch whose 128- or 256-row
 *                          tiling has <= 128 tiles lets up to 32 workgroups share a tile's K stages (at most 256 workgroups in all) whenever the engine's cost model
 *                          (fitted to tools/splitk_tile_sweep.py) predicts a gain of 5 % or more: each workgroup dumps its fp32 accumulators into a slab and a second
 *                          kernel adds the slabs in split order (deterministic) and runs the epilogue of the launch's mode.  40 MiB covers every launch (256 slabs of
 *                          160 KiB: 256-row tiles; 128-row tiles need half).  The workspace is scratch between a launch and its reduce kernel, both on the launch's
 *                          stream: launches on ONE stream may share it, launches on different streams need different workspaces.  max_splits: 0 = automatic, 1 = never
 *                          split, k > 1 = split every launch min(k, stages, 256 / tiles) ways (parity tests).  (NULL / 0 bytes: the launch runs unsplit)
 *   colstats_rows_per_block, colstats_blocks : out -- 0 / 0 when the launch wrote no column records (it split K, enumerated whole tensors, ...)
 *   rowstats_slots : out -- records per row written (0: none)
 *   splits         : out -- K splits of the launch (1: unsplit)
 *   flags          : in  -- 0, or GSW_MM_GN_ONLY (ABI 0.4.1; the word sits in what was tail padding: zero-initialise the struct): the PF output of this convolution
 *                           is read by nothing but a GroupNorm that takes its statistics from the column records requested here.  The launch then leaves the
 *                           one-pixel border of the output UNWRITTEN when (and only when) it wrote those records: the GroupNorm kernels never read it (the apply
 *                           kernel writes zeros to its own output's border whatever the input's holds) -- one small launch less per resnet. */
#define GSW_MM_GN_ONLY 1
typedef struct GswMmExtras {
    float* colstats_dev;
    int64_t colstats_capacity;
    float* rowstats_dev;
    int64_t rowstats_capacity;
    void* workspace_dev;
    int64_t workspace_bytes;
    int max_splits;
    int colstats_rows_per_block, colstats_blocks, rowstats_slots, splits;
    int flags;
} GswMmExtras;

/* gsw_gemm_strided / gsw_gemm_ln / gsw_conv_pf / gsw_conv3x3_res_pf / gsw_conv_up2x_pf with extras (ex may be NULL; the plain entry points ARE ex = NULL). */
int gsw_gemm_ex(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr, void* y_dev, int64_t ldy,
                int64_t M, int K, int N, int mode, int S, int Wimg, int dtype, GswMmExtras* ex, void* stream);
int gsw_gemm_ln_ex(const void* x_dev, const float* ln_stat_dev, const void* w_dev, const float* u_dev, const float* v_dev, void* y_dev, int64_t M, int K, int N,
                   int mode, int S, int dtype, GswMmExtras* ex, void* stream);
int gsw_conv_pf_ex(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                   int B, int H, int W, int C, int N, int ksize, int stride, int ldx, int dtype, GswMmExtras* ex, void* stream);
int gsw_conv3x3_res_pf_ex(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                          int B, int H, int W, int C, int N, const void* x1_dev, int C1, const void* x2_dev, int C2, int dtype, GswMmExtras* ex, void* stream);
int gsw_conv_up2x_pf_ex(const void* x_dev, const void* w4_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int N, int dtype, GswMmExtras* ex,
                        void* stream);

/* gsw_groupnorm_pf2 with the statistics folded from the column records of the producing launches (GswMmExtras.colstats_*; cs*_rows = rows per block the launch
 * reported, cs*_blocks = blocks per parity buffer, i.e. the buffer's stride): pixels per image must be a multiple of the block rows and the groups an even number
 * of channels wide; workspace_dev >= max(B * 64 * groups * 2, B * C) floats. */
int gsw_groupnorm_pf_cs(const void* x_dev, const void* x2_dev, int Ca, const float* cs1_dev, int cs1_rows, int cs1_npar, int cs1_blocks,
                        const float* cs2_dev, int cs2_rows, int cs2_npar, int cs2_blocks, const void* gamma_dev, const void* beta_dev, void* out_dev,
                        float* workspace_dev, int B, int H, int W, int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream);

/* ResnetBlock2D tail in ONE GEMM: y = conv3x3(x) + conv1x1([x1 | x2]) + bias + rowbias + resid on PF tensors.
 * w_dev: [N][9*C + C1 + C2] (3x3 taps first, then the shortcut's columns), x1 / x2 optional (x2 requires x1). */
int gsw_conv3x3_res_pf(const void* x_dev, const void* w_dev, const void* bias_dev, const void* rowbias_dev, int ld_rowbias, const void* resid_dev, void* y_dev,
                       int B, int H, int W, int C, int N, const void* x1_dev, int C1, const void* x2_dev, int C2, int dtype, void* stream);

/* Transformer blocks of the eps model: xnew = x + delta (skipped when delta_dev is NULL), y = LayerNorm(xnew) * gamma + beta;
 * x, delta, xnew, y: [rows, C]; C % 8 == 0, C <= 1536; GSW_F16 / GSW_BF16. */
int gsw_add_layernorm(const void* x_dev, const void* delta_dev, const void* gamma_dev, const void* beta_dev, void* xnew_dev, void* y_dev,
                      int64_t rows, int C, float eps, int dtype, void* stream);

/* Round-1 name of the dense linear layer: gsw_gemm(mode = geglu ? GSW_GEMM_GEGLU : GSW_GEMM_PLAIN). */

/* X2 / G1 -- every dense linear layer of the eps model (diffusers BasicTransformerBlock / Transformer2DModel / TimestepEmbedding /
 * ResnetBlock2D.time_emb_proj, which the reference reaches through `pipe(...)` at extract.py:66-69) on the hand-written matmul engine
 * (csrc/gswm_mm.hip: persistent 256 x 160 tiles, 8 waves in lockstep, three-stage LDS-DMA ring of 64-wide K slices):
 *   x: [M, K] row-major, w: [N, K] (nn.Linear layout), bias: [N] or NULL (PLAIN / GEGLU: 16-byte aligned -- the epilogue fetches it by LDS-DMA;
 *   GSW_ERR_BAD_ARG otherwise); K % 64 == 0, N % 8 == 0 (N % 160 == 0 for GEGLU; a partial last 160-column tile costs a full one); GSW_F16 / GSW_BF16.
 *   mode GSW_GEMM_PLAIN : y[M, N] = x w^T + bias (+ resid[M, N])
 *        GSW_GEMM_GEGLU : rows of w / bias interleaved per 16-ROW BLOCK as [8 value | 8 gate]: with I = N / 2 outputs, packed row
 *                         16 j + r holds value row 8 j + r for r < 8 and gate row I + 8 j + (r - 8) for r >= 8 of diffusers' [2 I, K]
 *                         projection (pf.pack_geglu_weight; bias packed the same way).  y[M, N/2] = value * gelu(gate) -- diffusers'
 *                         GEGLU without the [M, N] intermediate.  (ABI 0.1.x documented a per-160-row-tile [80 | 80] packing that the
 *                         engine never implemented since 0.2.0: GSW_VERSION >= 200 means the 16-row packing.)
 *        GSW_GEMM_TRANS : y[M/S][N][S] = per-image transpose of x w^T + bias (the attention kernel's V^T operand); S % 8 == 0
 *        GSW_GEMM_TOK2PF: rows are tokens (b, y, x) of S = H*W-token images of width Wimg; y_dev is row 0 of a padded-flat NHWC tensor
 *                         [B, H+2, W+2, N] and token rows land on its interior rows: y[pf(m)] = x w^T + bias (+ resid[pf(m)], which may
 *                         alias y: Transformer2DModel's `proj_out(...) + residual` in place) */
#define GSW_GEMM_PLAIN 0
#define GSW_GEMM_GEGLU 1
#define GSW_GEMM_TRANS 2
#define GSW_GEMM_TOK2PF 3
int gsw_gemm(const void* x_dev, const void* w_dev, const void* bias_dev, const void* resid_dev, void* y_dev, int64_t M, int K, int N,
             int mode, int S, int Wimg, int dtype, void* stream);
/* Self-attention's q | k | v projections from ONE pass over the tokens (diffusers Attention.to_q / to_k / to_v behind extract.py:66-69): w = the three
 * weights row-concatenated [N][K]; columns [0, N_rows) of x w^T (+ bias) -> rows_dev [M, N_rows] (q | k, read by gsw_attention as column slices),
 * columns [N_rows, N) -> trans_dev [M / S][N - N_rows][S] (V transposed).  N_rows % 160 == 0 (a column tile is either kind); K % 64, N % 8, S % 8 == 0. */
int gsw_gemm_qkv(const void* x_dev, const void* w_dev, const void* bias_dev, void* rows_dev, void* trans_dev, int64_t M, int K, int N_rows, int N,
                 int S, int dtype, void* stream);

/* LayerNorm folded into the GEMM that consumes it (diffusers BasicTransformerBlock: norm1 -> to_q | to_k | to_v, norm2 -> to_q, norm3 -> the GEGLU
 * projection): LN(x) W^T + b = rstd_m (x W'^T)_mn - rstd_m mean_m u_n + v_n with W' = W diag(gamma), u = W' 1, v = W beta + b -- the normalised tensor
 * is never written or read.
 *   row records          : GswMmExtras.rowstats_* of the launch that PRODUCES x (rowstats_slots records per row come back)
 *   gsw_ln_rowstats_finish: records -> stat_dev float2 [M] = (rstd, -rstd * mean) over the C columns.
 *   gsw_gemm_ln          : the consuming GEMM; w_dev = W' (GEGLU: packed like gsw_gemm), u_dev / v_dev fp32 [N] in the same row order, 16-byte aligned;
 *                          mode GSW_GEMM_PLAIN / GSW_GEMM_GEGLU / GSW_GEMM_TRANS; M % 8 == 0. */
int gsw_ln_rowstats_finish(const float* records_dev, int slots, int64_t M, int C, float eps, float* stat_dev, void* stream);
int gsw_gemm_ln(const void* x_dev, const float* ln_stat_dev, const void* w_dev, const float* u_dev, const float* v_dev, void* y_dev, int64_t M, int K, int N,
                int mode, int S, int dtype, void* stream);

/* The same with explicit row strides (elements, multiples of 8): x rows ldx >= K, w rows ldw >= K, resid rows ldr, y rows ldy -- operands may be
 * column slices of wider matrices (the per-image Q K^T and P V products of the VAE's single-head attention). */
int gsw_gemm_strided(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr,
                     void* y_dev, int64_t ldy, int64_t M, int K, int N, int mode, int S, int Wimg, int dtype, void* stream);

/* Process-wide tuning knobs of the matmul engine, for parity tests and A/B measurements only (production leaves both on "auto"; this is the
 * one piece of global state behind the ABI, also settable through the environment as GSW_MM_BM / GSW_MM_SPLIT before the first launch):
 *   tile_rows  : 0 = automatic (the 256 x 320 tile for long-K launches with several rounds of such tiles -- GSW_MM_WIDE is its per-epilogue bit mask --, else
 *                256 x 160 tiles unless they would leave CUs without one, then 128 x 160), 128 or 256 = forced narrow tiles, 512 = the 256 x 320 tile
 *                wherever it is legal (N % 320 == 0, dense rows / PF rows / GEGLU epilogues, operands below 4 GiB); -1 = keep
 *   split_mask : bit e set = epilogue kind e (0 dense rows -- 128-row tiles only: the 256-row dense-row tile has no 12-wave form --, 1 PF / convolution, 2 GEGLU, 3 transposed) runs the 12-wave variant whose
 *                waves 8-11 own the LDS-DMA; -1 = keep (default 10: convolutions and the transposed projection) */
int gsw_mm_config(int tile_rows, int split_mask);
int gsw_mm_get_config(int* tile_rows, int* split_mask); /* the current values (either pointer may be NULL): what a captured launch sequence depends on */

/* X1 / G1 tail -- diffusers AutoencoderKL mid-block attention (one head as wide as the block, 512): softmax over the rows of the score matrix
 * between the two engine products.  In place: x[r, 0:cols] <- softmax(scale * x[r, 0:cols]); rows `ld` elements apart; cols % 8 == 0. */
int gsw_softmax_rows(void* x_dev, int64_t rows, int cols, int64_t ld, float scale, int dtype, void* stream);

/* ---- The small-batch regime of the eps model (csrc/gswm_small.hip).  The reference inverts ONE latent per call (extract.py:112-117: 50 UNet
 * evaluations of one 4x64x64 latent), BASELINE configs[1] is batch 8: a forward is then a chain of ~500 dependent launches of 5-20 us, each paying
 * the ~4.7 us floor of a kernel boundary plus its own fill and drain, so what counts is FEWER and SHORTER launches.
 *
 * gsw_groupnorm_pf_fused: gsw_groupnorm_pf2 in ONE launch (no workspace): one workgroup per (image, group) holds the group's values in registers
 *   between the statistics and the normalisation (exact two-pass variance).  Groups an even number of channels wide; returns GSW_ERR_UNSUPPORTED
 *   when a group does not fit one workgroup's registers (more than 32 image rows per row lane, or an image row of the group wider than 1024 threads: use gsw_groupnorm_pf2 then) -- meant for B * groups
 *   up to a few hundred workgroups.
 * gsw_gather_rows: out[b] = table[clamp(index[b * index_stride], 0, nrows - 1)] (index_stride 0: one index for all rows): the time-embedding chain
 *   of diffusers' UNet2DConditionModel (sinusoid -> linear -> SiLU -> linear -> SiLU -> every resnet's time_emb_proj) depends on the timestep only,
 *   so it is tabulated once over the num_train_timesteps integer timesteps and a forward gathers its rows.  16-byte aligned rows.
 * gsw_nchw_to_pf: latent [B, Cin, H, W] -> PF rows [B, H+2, W+2, Cp]: zero border, channels >= Cin zero (conv_in reads one 64-wide K block).
 * gsw_conv3x3_pf_nchw: 3x3 stride-1 convolution of a PF tensor with Nout <= 16 output channels, written as NCHW [B, Nout, H, W] (+ bias): the
 *   UNet's conv_out (320 -> 4; diffusers UNet2DConditionModel.conv_out behind extract.py:66-69).  w: [Nout][9 * C] tap-major, channel-minor; C % 32 == 0. */
int gsw_groupnorm_pf_fused(const void* x_dev, const void* x2_dev, int Ca, const void* gamma_dev, const void* beta_dev, void* out_dev, int B, int H, int W,
                           int C, int groups, float eps, int act, int out_tokens, int dtype, void* stream);
int gsw_gather_rows(const void* table_dev, int64_t ld_bytes, int64_t nrows, const int64_t* index_dev, int index_stride, void* out_dev, int64_t out_ld_bytes,
                    int B, int64_t row_bytes, void* stream);
int gsw_nchw_to_pf(const void* x_dev, void* y_dev, int B, int Cin, int H, int W, int Cp, int dtype, void* stream);
int gsw_conv3x3_pf_nchw(const void* x_dev, const void* w_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int Nout, int dtype, void* stream);

/* Dense linears of the transformer blocks at SMALL M (one or two images: 64 ... 16384 token rows), where the engine's 128 x 160 tiles leave most CUs
 * without one: a four-wave workgroup owns a 64 x 64 / 32 x 64 / 16 x 64 / 16 x 32 output tile (config 0 .. 3; -1 = gsw_gemm_small_config's choice),
 * splits K four ways INSIDE the workgroup (fragments straight from global memory, partial tiles added in wave order through LDS: deterministic, no
 * second launch) and runs the epilogue of `mode` (GSW_GEMM_PLAIN / GEGLU / TRANS / TOK2PF, operands as for gsw_gemm_strided) with the engine's
 * rounding points.  K % 32 == 0, K >= 128, M % 16 == 0, N % 16 == 0 (GEGLU: N % 32 == 0), M <= 16384; 32-bit element offsets.
 *   ln_records_dev : LayerNorm fold (see gsw_gemm_ln) straight from the RAW row records [M][ln_slots][2] the producer of x left -- no finishing launch;
 *                    then w_dev = W diag(gamma), ln_u_dev / ln_v_dev fp32 [N] (16-byte aligned), bias_dev must be NULL; NULL = plain bias epilogue
 *   rowstats_dev   : GSW_GEMM_PLAIN only, optional: per output row and column tile the (sum, sum of squares) of the stored values,
 *                    [M][*rowstats_slots][2] floats, *rowstats_slots = ceil(N / tile columns) (0 when none were written);
 *                    capacity >= M * ceil(N / 32) * 2 covers every configuration.
 * gsw_gemm_small_config: the configuration the kernel would pick for a shape, or -1 when the shape is not this kernel's. */
int gsw_gemm_small_config(int64_t M, int K, int N, int mode);
int gsw_gemm_small(const void* x_dev, int64_t ldx, const void* w_dev, int64_t ldw, const void* bias_dev, const void* resid_dev, int64_t ldr, void* y_dev, int64_t ldy,
                   int64_t M, int K, int N, int mode, int S, int Wimg, const float* ln_records_dev, int ln_slots, float ln_eps, const float* ln_u_dev,
                   const float* ln_v_dev, float* rowstats_dev, int64_t rowstats_capacity, int* rowstats_slots, int config, int dtype, void* stream);

/* diffusers Upsample2D (nearest 2x + 3x3 convolution) from the low-resolution PF input, by sub-pixel decomposition: w4 =
 * [4 output parities (dy*2+dx)][N][4 taps (a*2+b)][C], the 3x3 weights pre-summed over the taps that read the same source pixel
 * (pf.pack_upsample_weight).  y: PF [B, 2H, 2W, N] (interior rows from the four parity launches, border rows zeroed here). */
int gsw_conv_up2x_pf(const void* x_dev, const void* w4_dev, const void* bias_dev, void* y_dev, int B, int H, int W, int C, int N, int dtype,
                     void* stream);

/* Attention of the eps-model (diffusers BasicTransformerBlock.attn1 / attn2 inside the UNet the reference runs at extract.py:66-69):
 * out = softmax(q k^T * scale) v per (batch, head), fp16 / bf16, fp32 accumulation -- a flash-attention forward.
 *   head_dim : 64 (SD 2.x), 40 / 80 / 160 (SD 1.x levels with 8 heads); other widths return GSW_ERR_UNSUPPORTED
 *   q   : [B, Sq, >= H*head_dim] row stride ldq elements, head h in columns [h*head_dim, (h+1)*head_dim)
 *   k   : [B, Sk, >= H*head_dim] row stride ldk
 *   vt  : [B, H*head_dim, Sk] contiguous -- the value projection TRANSPOSED (compute it as W_v x^T)
 *   out : [B, Sq, >= H*head_dim] row stride ldo
 *   Sk_valid : keys in [Sk_valid, Sk) are padding and get zero weight (cross-attention: 77 context tokens padded to 128)
 * Any Sq; Sk % 8 == 0; strides % 8 == 0; else GSW_ERR_UNSUPPORTED (sequences off the 128-query / 64-key tiles run a variant with clamped
 * loads and masked keys). */
int gsw_attention(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev, int B, int H, int head_dim, int Sq, int Sk,
                  int Sk_valid, int ldq, int ldk, int ldo, float scale, int dtype, void* stream);
/* The same with a caller-owned scratch buffer (16-byte aligned device memory, free again when the launch's stream has run it): few query tiles against many
 * key tiles -- one image's self-attention at 64 x 64 is 160 workgroups for 256 CUs, each walking 64 key tiles -- are then run as up to 8 workgroups per query
 * tile (at most 256 query tiles, at least 2048 keys), each over its own range of keys, and a second small kernel merges their partial results (deterministic).  4 MiB per 100 query tiles x splits is
 * enough (bytes = query tiles x splits x 4 x (16 ceil(head_dim / 32) + 2) x 256); a smaller buffer, or none, runs the unsplit kernel.  The split-K scratch of
 * GswMmExtras may be shared: both are free between launches of one stream. */
int gsw_attention_ws(const void* q_dev, const void* k_dev, const void* vt_dev, void* out_dev, int B, int H, int head_dim, int Sq, int Sk,
                     int Sk_valid, int ldq, int ldk, int ldo, float scale, int dtype, void* workspace_dev, int64_t workspace_bytes, void* stream);
/* The whole cross-attention sublayer of a transformer block as ONE launch (csrc/gswm_xattn.hip) -- diffusers' BasicTransformerBlock.attn2 with norm2 in front
 * and the residual behind, which extract.py:66-69 / the generation loop run at every latent level:
 *     out = x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(d)) V) + b_out,        K, V = to_k(ctx), to_v(ctx)
 * The context side is precomputed by the host ONCE per (context, layer) as a stream of MFMA fragments (xattn.py: context_operands; per head 112 640 bytes:
 * A' = K_h Wq_h diag(gamma) scale log2(e), rows centred, and B = Wo_h V_h^T (+ b_out as key slot 79 of the last head) in consumption order) plus 96 floats v per head
 * (the LayerNorm fold's constant term; -inf for padding keys).  x [x_images * tokens, 320] is the RAW residual stream, ln_stat float2 per row (rstd, -rstd mean:
 * gsw_ln_rowstats_finish); output image i reads x image i % x_images (classifier-free guidance: one x, two contexts) and context ctx_index[i] (NULL: context 0).
 * out_stat (nullable): (rstd, -rstd mean) of the OUTPUT rows with out_eps, what the next LayerNorm's fold reads.  C == 320, tokens % 128 == 0, <= 79 keys, else
 * GSW_ERR_UNSUPPORTED. */
int gsw_xattn_fused(const void* x_dev, const float* ln_stat_dev, const void* blob_dev, int64_t blob_stride_bytes, const float* v_dev, int64_t v_stride_floats,
                    const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images, int out_images, int tokens, int C, int heads,
                    int dtype, void* stream);
/* The same launch, starting one operation earlier in the block: it first MAKES x from the self-attention in front of the sublayer (BasicTransformerBlock.attn1's
 * output projection, bias and residual) -- x = resid + o Wo^T + b with o [x_images * tokens, 320] that attention's output and w_frag Wo and b as 21 chunks of
 * MFMA fragments (xattn.py: pack_out_projection) -- rounds it to the storage dtype as the separate launch would, takes norm2's statistics (epsilon ln_eps) from
 * the rounded rows, and continues as gsw_xattn_fused.  x is never written: the projection's own launch, its 2 x rows x 640 bytes of traffic and
 * gsw_ln_rowstats_finish all disappear.  Same shape limits. */
int gsw_xattn_fused_pre(const void* resid_dev, const void* o_dev, const void* w_frag_dev, float ln_eps, const void* blob_dev, int64_t blob_stride_bytes,
                        const float* v_dev, int64_t v_stride_floats, const int32_t* ctx_index_dev, void* out_dev, float* out_stat_dev, float out_eps, int x_images,
                        int out_images, int tokens, int C, int heads, int dtype, void* stream);

/* GroupNorm + proj_in of a transformer (diffusers' Transformer2DModel.forward: `self.norm(hidden_states)`, NHWC flatten, `self.proj_in`; behind extract.py:66-69) at the
 * 320-channel level as ONE launch: tokens[b, y W + x, :] = GroupNorm(x)[b, y, x, :] Wp^T + b, x a padded-flat NHWC tensor (gsw_conv_pf's layout), the rows normalised in
 * registers on their way into the matrix pipe -- the normalised tensor is never stored.  pairsum: per-image, per-column-PAIR (sum, sum of squares) of x's interior,
 * [B][C / 2][2] floats, from the producing launch's column records (gsw_gn_colstats_pairs: the first half of gsw_groupnorm_pf_cs on its own).  w_frag: Wp and b as 21
 * chunks of MFMA fragments (xattn.py: pack_out_projection).  out_stat (nullable): (rstd, -rstd mean) of the output rows with out_eps, for the LayerNorm that follows.
 * C == 320, W % 32 == 0, H W % 128 == 0, an even number of channels per group, else GSW_ERR_UNSUPPORTED. */
int gsw_gn_colstats_pairs(const float* cs_dev, int cs_rows, int cs_npar, int cs_blocks, float* pairsum_dev, int B, int H, int W, int C, void* stream);
int gsw_gn_proj_tokens(const void* x_pf_dev, const float* pairsum_dev, const void* gamma_dev, const void* beta_dev, float gn_eps, int groups, const void* w_frag_dev,
                       void* out_dev, float* out_stat_dev, float out_eps, int B, int H, int W, int C, int dtype, void* stream);

/* E4, bit-parity mode -- gs_insert.py:62 `np.random.uniform(0, 1)` / nodes.py:52-53,114-117 `RandomState(seed).uniform(0, 1)`:
 * NumPy's legacy MT19937 `random_sample` stream continued on the device.  key/pos: the generator state as
 * np.random.get_state()[1:3] gives it (pos == 624: block exhausted).  Writes n float64 uniforms to u_dev and, when
 * state_out_dev != NULL, the advanced state ([624] key words + [1] pos) so that the host generator can be moved past the draws.
 * gsw_mt19937_seed: NumPy's legacy integer seeding (init_genrand); use pos = 624 with it. */
void gsw_mt19937_seed(uint32_t seed, uint32_t key[624]);
int gsw_mt19937_uniform(const uint32_t key[624], int pos, double* u_dev, int64_t n, uint32_t* state_out_dev, void* stream);

/* =====================================================================================================================
 * Image-side stages either side of the latent loops.  Images are uint8 [B, H, W, 3]
 * (what np.asarray(PIL image) gives); every function restates, bit for bit, what Pillow / libjpeg compute for the call the
 * reference makes.
 * ===================================================================================================================== */
typedef enum gsw_image_mode {
    GSW_IMG_U8_HWC = 0,  /* uint8 [B, H, W, 3]: a PIL image                                                           */
    GSW_IMG_F16_CHW = 1, /* fp16 [B, 3, H, W] = fp16(2 * fp16(v / 255) - 1): extract.py:37,48,40 (ToTensor -> .to(float16)
                            -> img_to_latents' 2.*x - 1.), i.e. exactly what the VAE encoder is fed                     */
    GSW_IMG_F32_CHW = 2  /* fp32 [B, 3, H, W] = v / 255: torchvision ToTensor (extract.py:37)                          */
} gsw_image_mode;

/* extract.py:35-36 / distortions:229-230 `pil_img.resize(size, Image.Resampling.LANCZOS)`: Pillow's Resample.c
 * precompute_coeffs + normalize_coeffs_8bpc (double precision, host libm) for one axis.  Fills bounds[out_size*2] (first source
 * index, count) and kk[out_size*ksize] (22-bit fixed point); returns ksize > 0, or -gsw_status.  bounds == kk == NULL: returns
 * ksize only.  Pure host function (a "plan": upload both arrays once per (in_size, out_size)). */
int gsw_lanczos_plan(int in_size, int out_size, int32_t* bounds, int32_t* kk, int kk_capacity);

/* The two resampling passes of ImagingResample on device (horizontal first, uint8 in between; a pass whose size is unchanged is
 * skipped like Pillow does), with the conversion to `out_mode` fused into the last pass.  tmp_dev: [B, Hin, Wout, 3] uint8 scratch,
 * needed when a horizontal pass is followed by anything (may be NULL otherwise).  *_dev plan arrays: gsw_lanczos_plan output for
 * (Win -> Wout) and (Hin -> Hout); NULL for a skipped pass. */
int gsw_resize_lanczos(const uint8_t* in_dev, int B, int Hin, int Win, void* out_dev, int Hout, int Wout, int out_mode, uint8_t* tmp_dev,
                       const int32_t* hbounds_dev, const int32_t* hkk_dev, int hksize, const int32_t* vbounds_dev, const int32_t* vkk_dev,
                       int vksize, void* stream);

/* `decode_image` tail + diffusers numpy_to_pil (modified_stable_diffusion_gs.pyc src :207-220): [B, 3, H, W] float tensor ->
 * uint8 [B, H, W, 3] = (x * 255).round(); denormalise != 0 first applies (x / 2 + 0.5).clamp(0, 1) in the tensor's dtype. */
int gsw_tensor_to_image(const void* in_dev, int dtype, int B, int H, int W, int denormalise, uint8_t* out_dev, void* stream);

/* distortions:175-184 "compression": `image.save(buf, format="JPEG", quality=q)` then `Image.open(buf)`, minus the (lossless)
 * entropy coding: RGB -> YCbCr 4:2:0, islow FDCT, IJG quality-scaled tables (force_baseline), dequantise, islow IDCT, fancy h2v2
 * upsampling, YCbCr -> RGB.  workspace_dev: gsw_jpeg_workspace_bytes(B, H, W) bytes. */
int gsw_jpeg_quant_tables(int quality, uint8_t luma[64], uint8_t chroma[64]);
size_t gsw_jpeg_workspace_bytes(int B, int H, int W);
int gsw_jpeg_roundtrip(const uint8_t* rgb_dev, int B, int H, int W, int quality, void* out_dev, int out_mode, uint8_t* workspace_dev,
                       void* stream);

/* distortions:157-164 "blurring": `image.filter(ImageFilter.GaussianBlur(radius))` = Pillow's three extended-box passes per axis
 * (BoxBlur.c), bit-exact.  tmp_dev: uint8 scratch of the image batch's size.  gsw_gaussian_blur_params exposes the integer box
 * radius and the two 8.24 weights Pillow derives from `radius` (pure host). */
int gsw_gaussian_blur_params(float radius, int passes, int* box_radius, uint32_t* ww, uint32_t* fw);
int gsw_gaussian_blur(const uint8_t* rgb_dev, int B, int H, int W, float radius, uint8_t* out_dev, uint8_t* tmp_dev, void* stream);

/* Point-wise attacks of distortions:131-224. */
typedef enum gsw_pointwise_op {
    GSW_PW_BRIGHTNESS = 0, /* ImageEnhance.Brightness(image).enhance(strength)  (distortions:131-139)                  */
    GSW_PW_CONTRAST = 1,   /* ImageEnhance.Contrast(image).enhance(strength)    (distortions:141-148); workspace: [B] uint64 */
    GSW_PW_INVERT = 2,     /* F.invert                                          (distortions:222-223)                  */
    GSW_PW_GRAY = 3,       /* F.rgb_to_grayscale, replicated to 3 channels      (distortions:201-202)                  */
    GSW_PW_HFLIP = 4,      /* F.hflip                                           (distortions:203-204)                  */
    GSW_PW_VFLIP = 5,      /* F.vflip                                           (distortions:205-206)                  */
    GSW_PW_NOISE = 6       /* (x + strength * N(0,1)).clamp(0,1); Philox keyed by (seed; pixel, image_index0 + b) -- the
                              reference draws torch.randn on the host (distortions:166-173): statistical parity only      */
} gsw_pointwise_op;
int gsw_image_pointwise(const uint8_t* rgb_dev, int B, int H, int W, int op, float strength, uint64_t seed, uint64_t image_index0,
                        void* out_dev, int out_mode, uint64_t* workspace_dev, void* stream);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* GSWM_H */
