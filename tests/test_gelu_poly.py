"""The matmul engine's GELU (csrc/gswm_mmtypes.h: mm_gelu) is a clamped polynomial; its constants are read out of the header and the bound the header states is
re-derived in float32 Horner arithmetic against the erf form the reference's GEGLU computes (diffusers GEGLU.gelu -> F.gelu, reference extract.py:66-69 loads that UNet)."""
import os
import re

import numpy as np
from scipy.special import erf

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "a-watermark-for-diffusion-models_amd", "csrc", "gswm_mmtypes.h")


def _constants():
    src = open(HDR).read()
    clamp = float(re.search(r"#define MM_GELU_CLAMP ([0-9.]+)f", src).group(1))
    body = re.search(r"#define MM_GELU_COEFFS \{(.*?)\}", src, re.S).group(1)
    cf = [np.float32(v) for v in re.findall(r"-?[0-9]+\.[0-9eE+-]+(?=f)", body)]
    return np.float32(clamp), cf


def _mm_gelu(g32, clamp, cf):
    f = np.float32
    gm = np.maximum(g32, -clamp)
    c = np.minimum(gm, clamp)
    z = (c * c) * f(2.0 / float(clamp * clamp)) + f(-1.0)
    q = np.full_like(z, cf[0])
    for ck in cf[1:]:
        q = q * z + ck
    return gm * (c * q + f(0.5))


def test_polynomial_gelu_bound():
    clamp, cf = _constants()
    assert len(cf) == 13 and clamp == 5.0
    g = np.concatenate([np.linspace(-8, 8, 1600001), np.array([-65504.0, -1000.0, 1000.0, 65504.0])])
    ref = g * 0.5 * (1.0 + erf(g / np.sqrt(2.0)))
    out = _mm_gelu(g.astype(np.float32), clamp, cf).astype(np.float64)
    err = np.abs(out - ref)
    assert err[np.abs(g) <= 8].max() <= 2.5e-6                    # absolute, every finite fp16 gate in the range activations live in
    assert err[g < -8].max() <= 2.5e-6                            # huge negative gates: -5 Phi(-5) = -1.4e-6 instead of 0
    pos = g >= 0.01
    assert (err[pos] / ref[pos]).max() <= 2.5e-6                  # relative on the positive side (three orders below the fp16 rounding that follows)


def test_every_fp16_gate_rounds_like_the_erf_form_to_within_one_ulp():
    """The epilogue rounds gelu to fp16 right away: over ALL finite fp16 gates the rounded polynomial is the rounded erf form or its neighbour."""
    clamp, cf = _constants()
    bits = np.arange(0, 1 << 16, dtype=np.uint16)
    g16 = bits.view(np.float16)
    g16 = g16[np.isfinite(g16)]
    g = g16.astype(np.float64)
    ref16 = (g * 0.5 * (1.0 + erf(g / np.sqrt(2.0)))).astype(np.float16)
    out16 = _mm_gelu(g16.astype(np.float32), clamp, cf).astype(np.float16)
    # distance in fp16 steps through the monotone integer view
    def key(h):
        i = h.view(np.int16).astype(np.int32)
        return np.where(i < 0, -(i & 0x7FFF), i)
    steps = np.abs(key(out16) - key(ref16))
    big = np.abs(ref16.astype(np.float64)) >= 2.0 ** -9           # where one fp16 step is >= 1.9e-6
    assert steps[big].max() <= 1
    assert np.abs(out16.astype(np.float64) - ref16.astype(np.float64))[~big].max() <= 2.6e-6
