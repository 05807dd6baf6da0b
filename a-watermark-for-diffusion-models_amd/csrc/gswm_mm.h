// gswm_mm.h -- internal interface of the matmul engine (csrc/gswm_mm.hip), shared with the convolution front end (csrc/gswm_conv.hip).
#ifndef GSWM_MM_H
#define GSWM_MM_H
#include <stdint.h>

struct MMSeg {
    const void* x;       // A-operand rows of this K segment (row 0; PF tensors have guard rows at negative indices)
    int32_t ld;          // row stride in elements
    int32_t kblocks;     // channels / 64
    int32_t ntaps;       // 1, 4 (2 x 2) or 9 (3 x 3)
    int32_t tw;          // taps per tap row (1, 2, 3)
    int32_t tap_row;     // row offset between tap rows (= padded input width)
    int32_t tap_base;    // row offset of tap 0
    int32_t wk0;         // offset of this segment inside a weight row; inside it K runs tap-major, channel-minor
};

enum { MM_MODE_DENSE = 0, MM_MODE_PF = 1, MM_MODE_TOK2PF = 2, MM_MODE_UP2X = 3, MM_MODE_GEGLU = 4, MM_MODE_TRANS = 5, MM_MODE_QKV = 6 };
enum { MM_FLAG_NONE = 0, MM_FLAG_COMPACT = 1 };   // COMPACT (MM_MODE_PF / UP2X): M enumerates interior pixels, borders are not written

struct MMArgs {
    MMSeg seg[3];
    int32_t nseg;
    int32_t P;            // stages (K / 64 summed over segments and taps) per tile
    const void* w;        // [N][ldw]
    int32_t ldw;
    int32_t M, N;         // M: rows of the output row space (all padded-flat rows, or the interior pixels with MM_FLAG_COMPACT)
    int32_t tiles_n, ntiles;      // filled by gsw_mm_launch
    int32_t panel;                // filled by gsw_mm_launch: column tiles per panel of the tile order (8; all of them when their weight tiles fit an XCD's L2)
    const void* bias;     // [N] or null
    const void* rowbias;  // [images][ldrb] or null (MM_MODE_PF): per-image row bias, rows ldrb elements apart (a column slice of a wider matrix)
    const void* resid;    // [rows][ldr] or null, addressed like the output
    void* y;
    void* y2;             // MM_MODE_QKV: the transposed part [images][N - n_rows][S] (columns >= n_rows); y takes columns < n_rows as dense rows
    int32_t n_rows;       // MM_MODE_QKV: columns of the row-major part (a multiple of the 160-column tile)
    int32_t ldy, ldr;
    int32_t ldrb;         // row stride of rowbias (>= N, multiple of 8)
    int32_t mode;
    int32_t Hp, Wp;       // padded geometry of the output row space (PF / UP2X) or of the target PF tensor (TOK2PF)
    int32_t in_Hp, in_Wp; // MM_FLAG_COMPACT: padded geometry of the INPUT tensor (== Hp, Wp unless stride 2)
    int32_t stride;       // MM_FLAG_COMPACT: 1 or 2 -- output pixel (y, x) reads input rows around (stride y, stride x)
    int32_t S, Wimg;      // tokens per image (TRANS, TOK2PF), image width (TOK2PF)
    int32_t up;           // UP2X: 1 + dy * 2 + dx
    int32_t flags;        // MM_FLAG_*
    int32_t splits;       // filled by gsw_mm_launch: > 1 = split-K (the K stages of a tile are shared by `splits` workgroups, fp32 partials in ws)
    float* colstats;      // EPI 1 only, or null: per 16 MT-row block and output column PAIR the sum and the sum of squares of the STORED values,
                          // [blocks = 4 tiles_m][2 planes][N / 2] floats -- GroupNorm statistics of the output without another pass over it
    // LayerNorm folded into the GEMM (LNF kernels; dense / GEGLU / transposed modes): y = rstd_m (x W'^T)_mn + nrm_m u_n + v_n with W' = W diag(gamma),
    // u = W' 1, v = W beta + bias, (rstd_m, nrm_m = -rstd_m mean_m) per row of x
    const void* ln_stat;  // float2 [M] or null
    const float* ln_u;    // [N]
    const float* ln_v;    // [N]
    float* rowstats;      // EPI 0 only, or null: per output row and 80-column half tile the (sum, sum of squares) of the STORED values, [M][2 tiles_n][2] floats
    float* ws;            // filled by gsw_mm_launch: split-K workspace, [splits][ntiles][8 waves][5 * MT accumulators][64 lanes] float4
};

// ex: the launch's extras (records requested, split-K scratch; results written back).  nullptr = none (gsw_mm_no_extras).
struct GswMmExtras;
int gsw_mm_launch(MMArgs& a, int dtype, void* stream, GswMmExtras* ex);
// microseconds the engine's plan (tiling, split-K) predicts for M x N outputs over P stages with these extras (nullptr: none): the convolution
// front end chooses its row enumeration with it
double gsw_mm_predict_us(int64_t M, int N, int P, const GswMmExtras* ex);
// an extras struct that requests nothing
void gsw_mm_no_extras(GswMmExtras* ex);

#endif
