// gswm_mmtypes.h -- device-side element helpers shared by the matmul engine (gswm_mm.hip) and the small-batch kernels (gswm_small.hip): MFMA fragment types,
// conversions with the storage dtype's rounding, packed adds / multiplies, the GEGLU epilogue's gelu.  gfx950 only.
#ifndef GSWM_MMTYPES_H
#define GSWM_MMTYPES_H
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

namespace {

typedef _Float16 mm_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 mm_b8 __attribute__((ext_vector_type(8)));
typedef float mm_f4 __attribute__((ext_vector_type(4)));

template <typename T> struct MM;
template <> struct MM<_Float16> {
    typedef mm_h8 frag;
    static __device__ __forceinline__ mm_f4 mma(frag a, frag b, mm_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) { return __half_as_ushort(__float2half_rn(f)); }
    static __device__ __forceinline__ float up(uint16_t h) { return __half2float(__ushort_as_half(h)); }
    static __device__ __forceinline__ uint32_t cvt2(float lo, float hi) {      // v_cvt_pk_f16_f32 (round to nearest even)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, h2));
    }
    static __device__ __forceinline__ float up_lo(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[0]; }
    static __device__ __forceinline__ float up_hi(uint32_t p) { typedef _Float16 h2 __attribute__((ext_vector_type(2))); return (float)__builtin_bit_cast(h2, p)[1]; }
    static __device__ __forceinline__ uint32_t mul2(uint32_t a, uint32_t b) {  // v_pk_mul_f16: the correctly rounded fp16 product of fp16 operands, two at a time
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, a) * __builtin_bit_cast(h2, b));
    }
    static __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) {  // v_pk_add_f16: the correctly rounded sum of two fp16 values = cvt(up(a) + up(b)), two at a time
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, a) + __builtin_bit_cast(h2, b));
    }
    // (sum, sum of squares) of a packed pair into fp32 accumulators: two v_dot2_f32_f16 (exact products, fp32 accumulation)
    static __device__ __forceinline__ void stat2(uint32_t w, float& sm, float& sq) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        const h2 v = __builtin_bit_cast(h2, w);
        sm = __builtin_amdgcn_fdot2(v, h2{(_Float16)1.0f, (_Float16)1.0f}, sm, false);
        sq = __builtin_amdgcn_fdot2(v, v, sq, false);
    }
};
template <> struct MM<__bf16> {
    typedef mm_b8 frag;
    static __device__ __forceinline__ mm_f4 mma(frag a, frag b, mm_f4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ uint16_t cvt(float f) {
        union { __hip_bfloat16 h; uint16_t u; } c; c.h = __float2bfloat16(f); return c.u;
    }
    static __device__ __forceinline__ float up(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
    static __device__ __forceinline__ uint32_t cvt2(float lo, float hi) {      // v_cvt_pk_bf16_f32 (round to nearest even)
        typedef __bf16 b2 __attribute__((ext_vector_type(2)));
        typedef float f2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
    }
    static __device__ __forceinline__ float up_lo(uint32_t p) { return __uint_as_float(p << 16); }
    static __device__ __forceinline__ float up_hi(uint32_t p) { return __uint_as_float(p & 0xFFFF0000u); }
    static __device__ __forceinline__ uint32_t mul2(uint32_t a, uint32_t b) {  // bf16 x bf16 is exact in fp32: one rounding, like a bf16 multiply
        return cvt2(up_lo(a) * up_lo(b), up_hi(a) * up_hi(b));
    }
    static __device__ __forceinline__ uint32_t add2(uint32_t a, uint32_t b) { return cvt2(up_lo(a) + up_lo(b), up_hi(a) + up_hi(b)); }
    static __device__ __forceinline__ void stat2(uint32_t w, float& sm, float& sq) {
        const float lo = up_lo(w), hi = up_hi(w);
        sm += lo + hi;
        sq = fmaf(lo, lo, fmaf(hi, hi, sq));
    }
};

// Abramowitz & Stegun 7.1.26 for the GEGLU epilogue: erfc(x) = (a1 t + ... + a5 t^5) exp(-x^2), t = 1 / (1 + p x), |error| <= 1.5e-7 (three orders below the
// fp16 / bf16 rounding of the gelu it feeds), branch-free.  The library erff (two polynomial branches, both executed in a wave) made the epilogue of the
// L0 feed-forward projection (K = 320: five K slices per tile) VALU-bound.
// gelu(g) = g Phi(g) for the GEGLU epilogue, from h = erfc(|g| / sqrt 2) / 2 (the same 7.1.26 polynomial with the 1/2 and the 1/sqrt 2 folded into
// its constants): gelu = max(g, 0) - |g| h -- no sign transfer, no 1 + erf, 13 instructions with the v_rcp and the v_exp.
__device__ __forceinline__ float mm_gelu(float g) {
    const float ag = fabsf(g);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ag, 1.0f));
    float pl = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    pl = fmaf(pl, t, 0.5f * 1.421413741f);
    pl = fmaf(pl, t, 0.5f * -0.284496736f);
    pl = fmaf(pl, t, 0.5f * 0.254829592f);
    const float y = 0.84932180028801904f * g;                          // sqrt(log2(e) / 2) g: exp(-g^2 / 2) = exp2(-y^2)
    const float e = __builtin_amdgcn_exp2f(-y * y);
    return fmaf(-ag, pl * t * e, fmaxf(g, 0.0f));
}

}  // namespace

#endif
