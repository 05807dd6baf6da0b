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

// gelu(g) = g Phi(g) for the GEGLU epilogue WITHOUT transcendental instructions: Phi(g) = 1/2 + c Q(z), c = g clamped to [-5, 5], z = 2 c^2 / 25 - 1, Q a degree-12
// polynomial fitted for the smallest maximum of |g| |Phi_fit - Phi| (tools/gelu_poly_fit.py prints these coefficients and the bounds: |error| <= 1.7e-6 absolute in
// float32 Horner arithmetic for every g, <= 1.5e-6 relative for g > 0.01 -- the result is rounded to fp16 / bf16 next, eps 4.9e-4; tests/test_gelu_poly.py re-derives the
// bound from the constants below).  Beyond the clamp Phi stays Phi(+-5) = 1 - 2.9e-7 / 2.9e-7, and the negative side multiplies max(g, -5), so a huge negative gate
// gives -1.4e-6 instead of 0.  Every operation is an FMA / multiply / min / max: hipcc packs two values per v_pk_fma_f32, 40 issue cycles per value on a SIMD.
// (Rounds 1-4: Abramowitz & Stegun 7.1.26 with a v_rcp_f32 and a v_exp_f32 -- quarter-rate instructions, 16 cycles each per wave: 66 cycles per value, and the
// wide tile's GEGLU epilogue (80 values per lane) measured VALU-bound at 43 % of a K = 320 tile, profiles/r05m_mm_trace_wide.txt.)
#define MM_GELU_CLAMP 5.0f
#define MM_GELU_COEFFS { 0.0004929697024635971f, -0.001514154253527522f, 0.0020018599461764097f, -0.0029156720265746117f, 0.006165430881083012f, \
                         -0.010973232798278332f, 0.0164976567029953f, -0.023336730897426605f, 0.03142622858285904f, -0.040432948619127274f, \
                         0.0515214204788208f, -0.07029665261507034f, 0.14136378467082977f }      /* highest degree first */
__device__ __forceinline__ float mm_gelu(float g) {
    constexpr float cf[13] = MM_GELU_COEFFS;
    const float gm = fmaxf(g, -MM_GELU_CLAMP), c = fminf(gm, MM_GELU_CLAMP);
    const float z = fmaf(c * c, 2.0f / (MM_GELU_CLAMP * MM_GELU_CLAMP), -1.0f);
    float q = cf[0];
#pragma unroll
    for (int i = 1; i < 13; ++i) q = fmaf(q, z, cf[i]);
    return gm * fmaf(c, q, 0.5f);
}

}  // namespace

#endif
