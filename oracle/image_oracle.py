"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy / integer arithmetic) of the image stages either side of the latent loops
(SURVEY.md section 8f ranks 1-2).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

The reference does these stages through third-party libraries that are not vendored in /root/reference:

* extract.py:31-37 `load_image` and distortions:226-233 ("scaling"): `PIL.Image.resize(size, Image.Resampling.LANCZOS)`  -> Pillow's
  two-pass fixed-point resampler (src/libImaging/Resample.c: precompute_coeffs / normalize_coeffs_8bpc /
  ImagingResampleHorizontal_8bpc / ImagingResampleVertical_8bpc, 22 fractional bits, uint8 intermediate).
* extract.py:37 `tvt.ToTensor()`, :48 `.to(dtype=float16)`, :40 `2. * x - 1.`                             -> `normalise_like_reference`.
* distortions:175-184 ("compression"): `image.save(buf, format="JPEG", quality=q)` / `Image.open(buf)`     -> libjpeg(-turbo) with
  Pillow's defaults: YCbCr 4:2:0 (h2v2 box downsample), integer "islow" forward DCT, IJG quality-scaled Annex-K tables
  (force_baseline), then the decoder's islow inverse DCT, "fancy" (triangle) h2v2 chroma upsampling and fixed-point YCbCr->RGB.
  Entropy coding is lossless and therefore not part of the distortion.
* distortions:131-155 brightness / contrast (`ImageEnhance` = `Image.blend` with a black / mean-grey image), :203-204,224 togray,
  flips, invert.

Requirements pin none of Pillow / libjpeg (requirements.txt lists diffusers, cryptography, scipy, tqdm, typing), so parity is
pinned against the Pillow that IS importable (12.2.0 with libjpeg-turbo here and on the GPU box): tests/test_image_oracle.py checks
every function below bit-for-bit against PIL on seeded images.
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c


# ---------------------------------------------------------------------------------------------------------------------
# Pillow Lanczos resampling (uint8, any number of interleaved channels)
# ---------------------------------------------------------------------------------------------------------------------
def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3.0)
    return 0.0


def lanczos_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box: (bounds[out,2], kk[out,ksize] int32, ksize)."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 3.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One pass along `axis` of an [H, W, C] uint8 image; skipped when the size is unchanged (ImagingResample: need_horizontal /
    need_vertical)."""
    if img.shape[axis] == out_size:
        return img
    bounds, kk, _ = lanczos_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.tensordot(kk[xx, :xmax].astype(np.int64), src[xmin:xmin + xmax], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize_lanczos(img: np.ndarray, size: Tuple[int, int]) -> np.ndarray:
    """`PIL.Image.fromarray(img).resize(size, LANCZOS)` for an [H, W, C] uint8 array; size = (width, height) as PIL takes it.
    Horizontal pass first, uint8 in between (Resample.c ImagingResample)."""
    w, h = size
    return _resample_axis(_resample_axis(np.ascontiguousarray(img), w, 1), h, 0)


def normalise_like_reference(img_u8: np.ndarray) -> np.ndarray:
    """extract.py:37,48,40: ToTensor (float32 v/255, CHW) -> .to(float16) -> 2.*x - 1. evaluated in float16."""
    x = (img_u8.astype(np.float32) / np.float32(255.0)).astype(np.float16)
    y = ((x * np.float16(2.0)).astype(np.float16) - np.float16(1.0)).astype(np.float16)
    return np.ascontiguousarray(np.moveaxis(y, -1, 0))


# ---------------------------------------------------------------------------------------------------------------------
# JPEG lossy stages (libjpeg / libjpeg-turbo, Pillow defaults)
# ---------------------------------------------------------------------------------------------------------------------
STD_LUMA_Q = np.array([
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99],
    dtype=np.int64).reshape(8, 8)
STD_CHROMA_Q = np.array([
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99],
    dtype=np.int64).reshape(8, 8)


def jpeg_quant_tables(quality: int) -> Tuple[np.ndarray, np.ndarray]:
    """jcparam.c jpeg_quality_scaling + jpeg_add_quant_table with force_baseline (Pillow passes TRUE)."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for base in (STD_LUMA_Q, STD_CHROMA_Q):
        t = (base * scale + 50) // 100
        out.append(np.clip(t, 1, 255))
    return out[0], out[1]


def rgb_to_ycc(rgb: np.ndarray):
    """jccolor.c rgb_ycc_convert (16-bit fixed point)."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    return y, cb, cr


def _pad_edge(p: np.ndarray, mh: int, mw: int) -> np.ndarray:
    H, W = p.shape
    return np.pad(p, ((0, (-H) % mh), (0, (-W) % mw)), mode="edge")


def h2v2_downsample(p: np.ndarray) -> np.ndarray:
    """jcsample.c h2v2_downsample: 2x2 box with the alternating 1,2 rounding bias along a row."""
    a = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
    bias = np.where(np.arange(a.shape[1]) % 2 == 0, 1, 2)[None, :]
    return (a + bias) >> 2


_CB, _PB = 13, 2          # CONST_BITS, PASS1_BITS (8-bit samples)
_F = dict(c0298=2446, c0390=3196, c0541=4433, c0765=6270, c0899=7373, c1175=9633, c1501=12299, c1847=15137, c1961=16069, c2053=16819,
          c2562=20995, c3072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def fdct_islow(block: np.ndarray) -> np.ndarray:
    """jfdctint.c jpeg_fdct_islow on [..., 8, 8] int64 samples (already level-shifted by -128); output scaled by 8."""
    F = _F
    d = block.astype(np.int64)

    def pass_(d, first):
        d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
        tmp0, tmp7 = d0 + d7, d0 - d7
        tmp1, tmp6 = d1 + d6, d1 - d6
        tmp2, tmp5 = d2 + d5, d2 - d5
        tmp3, tmp4 = d3 + d4, d3 - d4
        tmp10, tmp13 = tmp0 + tmp3, tmp0 - tmp3
        tmp11, tmp12 = tmp1 + tmp2, tmp1 - tmp2
        o = [None] * 8
        if first:
            o[0] = (tmp10 + tmp11) << _PB
            o[4] = (tmp10 - tmp11) << _PB
            sh = _CB - _PB
        else:
            o[0] = _descale(tmp10 + tmp11, _PB)
            o[4] = _descale(tmp10 - tmp11, _PB)
            sh = _CB + _PB
        z1 = (tmp12 + tmp13) * F["c0541"]
        o[2] = _descale(z1 + tmp13 * F["c0765"], sh)
        o[6] = _descale(z1 + tmp12 * (-F["c1847"]), sh)
        z1, z2, z3, z4 = tmp4 + tmp7, tmp5 + tmp6, tmp4 + tmp6, tmp5 + tmp7
        z5 = (z3 + z4) * F["c1175"]
        tmp4 = tmp4 * F["c0298"]
        tmp5 = tmp5 * F["c2053"]
        tmp6 = tmp6 * F["c3072"]
        tmp7 = tmp7 * F["c1501"]
        z1 = z1 * (-F["c0899"])
        z2 = z2 * (-F["c2562"])
        z3 = z3 * (-F["c1961"]) + z5
        z4 = z4 * (-F["c0390"]) + z5
        o[7] = _descale(tmp4 + z1 + z3, sh)
        o[5] = _descale(tmp5 + z2 + z4, sh)
        o[3] = _descale(tmp6 + z2 + z3, sh)
        o[1] = _descale(tmp7 + z1 + z4, sh)
        return np.stack(o, axis=-1)

    rows = pass_(d, True)                                        # pass 1: along each row (last axis)
    cols = pass_(np.swapaxes(rows, -1, -2), False)               # pass 2: along each column
    return np.swapaxes(cols, -1, -2)


def quantize(coef: np.ndarray, q: np.ndarray) -> np.ndarray:
    """jcdctmgr.c quantize: round-half-away of coef / (8 q) (the forward DCT output carries a factor 8)."""
    qv = q.astype(np.int64) << 3
    a = np.abs(coef) + (qv >> 1)
    return np.sign(coef) * (a // qv)


def idct_islow(coef: np.ndarray) -> np.ndarray:
    """jidctint.c jpeg_idct_islow on dequantised [..., 8, 8] coefficients -> uint8 samples (range-limited, +128)."""
    F = _F
    c = coef.astype(np.int64)

    def pass_(c, first):
        i0, i1, i2, i3, i4, i5, i6, i7 = (c[..., i] for i in range(8))
        z2, z3 = i2, i6
        z1 = (z2 + z3) * F["c0541"]
        tmp2 = z1 + z3 * (-F["c1847"])
        tmp3 = z1 + z2 * F["c0765"]
        tmp0 = (i0 + i4) << _CB
        tmp1 = (i0 - i4) << _CB
        tmp10, tmp13 = tmp0 + tmp3, tmp0 - tmp3
        tmp11, tmp12 = tmp1 + tmp2, tmp1 - tmp2
        t0, t1, t2, t3 = i7, i5, i3, i1
        z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
        z5 = (z3 + z4) * F["c1175"]
        t0 = t0 * F["c0298"]
        t1 = t1 * F["c2053"]
        t2 = t2 * F["c3072"]
        t3 = t3 * F["c1501"]
        z1 = z1 * (-F["c0899"])
        z2 = z2 * (-F["c2562"])
        z3 = z3 * (-F["c1961"]) + z5
        z4 = z4 * (-F["c0390"]) + z5
        t0 = t0 + z1 + z3
        t1 = t1 + z2 + z4
        t2 = t2 + z2 + z3
        t3 = t3 + z1 + z4
        sh = _CB - _PB if first else _CB + _PB + 3
        o = [_descale(tmp10 + t3, sh), _descale(tmp11 + t2, sh), _descale(tmp12 + t1, sh), _descale(tmp13 + t0, sh),
             _descale(tmp13 - t0, sh), _descale(tmp12 - t1, sh), _descale(tmp11 - t2, sh), _descale(tmp10 - t3, sh)]
        return np.stack(o, axis=-1)

    cols = pass_(np.swapaxes(c, -1, -2), True)                   # pass 1: columns
    rows = pass_(np.swapaxes(cols, -1, -2), False)               # pass 2: rows
    return np.clip(rows + 128, 0, 255)


def _blocks(p: np.ndarray) -> np.ndarray:
    H, W = p.shape
    return p.reshape(H // 8, 8, W // 8, 8).swapaxes(1, 2)


def _unblocks(b: np.ndarray) -> np.ndarray:
    nh, nw = b.shape[:2]
    return b.swapaxes(1, 2).reshape(nh * 8, nw * 8)


def _codec_plane(p: np.ndarray, q: np.ndarray) -> np.ndarray:
    """FDCT -> quantise -> dequantise -> IDCT of one padded component plane."""
    co = quantize(fdct_islow(_blocks(p) - 128), q)
    return _unblocks(idct_islow(co * q))


def h2v2_fancy_upsample(c: np.ndarray) -> np.ndarray:
    """jdsample.c h2v2_fancy_upsample: triangle filter, 3/4 nearer + 1/4 further in each axis; the context rows beyond the image
    are the edge rows themselves (jdmainct.c context wraparound at the top / bottom of the image)."""
    c = c.astype(np.int64)
    Hc, Wc = c.shape
    up = np.pad(c, ((1, 1), (0, 0)), mode="edge")
    out = np.empty((2 * Hc, 2 * Wc), dtype=np.int64)
    for v in (0, 1):
        near = up[1:-1]
        far = up[0:-2] if v == 0 else up[2:]
        colsum = 3 * near + far                                      # [Hc, Wc]
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], axis=1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], axis=1)
        even = (colsum * 3 + last + 8) >> 4
        odd = (colsum * 3 + nxt + 7) >> 4
        even[:, 0] = (colsum[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (colsum[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out


def ycc_to_rgb(y, cb, cr) -> np.ndarray:
    """jdcolor.c ycc_rgb_convert tables."""
    cbx, crx = cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    r = y + ((91881 * crx + 32768) >> 16)
    g = y + ((-22554 * cbx + 32768 - 46802 * crx) >> 16)
    b = y + ((116130 * cbx + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def jpeg_roundtrip(rgb: np.ndarray, quality: int) -> np.ndarray:
    """[H, W, 3] uint8 -> the image `PIL.Image.open(BytesIO(save(format="JPEG", quality=q)))` decodes to."""
    H, W, _ = rgb.shape
    ql, qc = jpeg_quant_tables(quality)
    y, cb, cr = rgb_to_ycc(rgb)
    yp = _pad_edge(y, 16, 16)
    # chroma: columns of the full-resolution plane are edge-padded to the MCU grid (jcsample.c expand_right_edge), rows only to one
    # row group (2 rows, jcprepct.c pre_process_data); the box-downsampled plane is then padded to the iMCU height by replicating
    # its own last row (expand_bottom_edge on the output buffer)
    cbp, crp = (_pad_edge(h2v2_downsample(_pad_edge(c, 2, 16)), 8, 8) for c in (cb, cr))
    yr = _codec_plane(yp, ql)
    cbr = _codec_plane(cbp, qc)
    crr = _codec_plane(crp, qc)
    # the decoder upsamples only the downsampled image area ceil(W/2) x ceil(H/2)
    hc, wc = (H + 1) // 2, (W + 1) // 2
    # jdsample.c jinit_upsampler: the triangle filter is only selected when downsampled_width > 2, else plain 2x2 replication
    up = h2v2_fancy_upsample if wc > 2 else (lambda c: np.repeat(np.repeat(c, 2, axis=0), 2, axis=1))
    cbu = up(cbr[:hc, :wc])[:H, :W]
    cru = up(crr[:hc, :wc])[:H, :W]
    return ycc_to_rgb(yr[:H, :W], cbu, cru)


# ---------------------------------------------------------------------------------------------------------------------
# point-wise distortions (PIL ImageEnhance / ImageOps restated)
# ---------------------------------------------------------------------------------------------------------------------
def rgb_to_l(rgb: np.ndarray) -> np.ndarray:
    """Convert.c rgb2l: L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16."""
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(a: np.ndarray, b: np.ndarray, alpha: float) -> np.ndarray:
    """Blend.c ImagingBlend: a + alpha (b - a) in C float arithmetic, truncated; clipped when alpha is outside [0, 1]."""
    al = np.float32(alpha)
    t = a.astype(np.float32) + al * (b.astype(np.float32) - a.astype(np.float32))
    if 0.0 <= alpha <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t.astype(np.int32))).astype(np.uint8)


def enhance_brightness(rgb: np.ndarray, factor: float) -> np.ndarray:
    """ImageEnhance.Brightness(image).enhance(factor) (distortions:131-139)."""
    return blend_u8(np.zeros_like(rgb), rgb, factor)


def enhance_contrast(rgb: np.ndarray, factor: float) -> np.ndarray:
    """ImageEnhance.Contrast(image).enhance(factor) (distortions:141-148): blend with the rounded mean grey level."""
    hist_mean = rgb_to_l(rgb).astype(np.float64).mean()
    mean = int(hist_mean + 0.5)
    return blend_u8(np.full_like(rgb, mean), rgb, factor)


# ---------------------------------------------------------------------------------------------------------------------
# Gaussian blur (PIL ImageFilter.GaussianBlur = three passes of an extended box blur per axis; BoxBlur.c)
# ---------------------------------------------------------------------------------------------------------------------
def gaussian_box_radius(radius: float, passes: int = 3) -> np.float32:
    """BoxBlur.c _gaussian_blur_radius: float / double mix exactly as the C expression evaluates."""
    f32 = np.float32
    sigma2 = f32(f32(f32(radius) * f32(radius)) / f32(passes))
    L = f32(np.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(np.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(f32(2) * l + f32(1)) * f32(f32(l * f32(l + f32(1))) - f32(f32(3) * sigma2)))
    a = f32(a / f32(f32(6) * f32(sigma2 - f32(f32(l + f32(1)) * f32(l + f32(1))))))
    return f32(l + a)


def box_blur_params(float_radius) -> Tuple[int, int, int]:
    """ImagingHorizontalBoxBlur: integer radius and the 8.24 fixed-point weights of the window body (ww) and its two fractional ends (fw)."""
    fr = np.float32(float_radius)
    radius = int(fr)
    ww = int(np.float32(16777216.0) / np.float32(fr * np.float32(2) + np.float32(1)))
    fw = ((1 << 24) - (radius * 2 + 1) * ww) // 2
    return radius, ww, fw


def _box_blur_axis(img: np.ndarray, radius: int, ww: int, fw: int, axis: int) -> np.ndarray:
    """One extended-box pass along `axis` of [H, W, C] uint8: (ww * sum(window) + fw * (left + right) + 2^23) >> 24, edges replicated."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    n = src.shape[0]
    idx = np.arange(n)
    acc = np.zeros_like(src)
    for d in range(-radius, radius + 1):
        acc += src[np.clip(idx + d, 0, n - 1)]
    far = src[np.clip(idx - radius - 1, 0, n - 1)] + src[np.clip(idx + radius + 1, 0, n - 1)]
    out = ((acc * ww + far * fw + (1 << 23)) & 0xFFFFFFFF) >> 24
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def gaussian_blur(img: np.ndarray, radius: float, passes: int = 3) -> np.ndarray:
    """`PIL.Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius))` (distortions:157-164) for [H, W, 3] uint8."""
    if radius == 0:
        return img.copy()
    r, ww, fw = box_blur_params(gaussian_box_radius(radius, passes))
    out = img
    for _ in range(passes):
        out = _box_blur_axis(out, r, ww, fw, 1)
    for _ in range(passes):
        out = _box_blur_axis(out, r, ww, fw, 0)
    return out
