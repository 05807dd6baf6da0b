"""Image-side stages either side of the latent loops, on the device (csrc/gswm_image.hip).

Host-side mirror of what the reference does with PIL / torchvision on the CPU, one image at a time:

* `load_image` of extract.py:31-37 (Lanczos resize -> ToTensor) plus the `.to(float16)` / `2.*x - 1.` that follow (extract.py:48,40)
  -> `resize_lanczos(..., out="f16")`: the fp16 CHW batch the VAE encoder consumes, bit-identical to the reference's tensor.
* `decode_image` / numpy_to_pil of the generation pipeline -> `tensor_to_image`.
* the `distortions` tool (apply_single_distortion): "compression" (JPEG QF), "scaling", "blurring", "brightness", "contrast", "noise",
  "togray", "invert", "horizontal_flip", "vertical_flip" -> `apply_distortion`, same strength conventions (`relative_strength_to_absolute`).

Images are uint8 [B, H, W, 3] device tensors (np.asarray(PIL image) stacked).  There is no CPU fallback: every function launches the
HIP kernels of libgswm through the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch

from . import _native as N
from .codec import _dt, _stream_ptr

_MODES = {"u8": N.GSW_IMG_U8_HWC, "f16": N.GSW_IMG_F16_CHW, "f32": N.GSW_IMG_F32_CHW}
_PLANS: Dict[Tuple[int, int, str], Tuple[torch.Tensor, torch.Tensor, int]] = {}


def _check_images(images: torch.Tensor):
    if images.dtype != torch.uint8 or images.dim() != 4 or images.shape[-1] != 3 or not images.is_cuda:
        raise ValueError("images must be a uint8 [B, H, W, 3] device tensor")
    return images.contiguous()


def _alloc_out(B: int, H: int, W: int, out: str, device) -> torch.Tensor:
    if out == "u8":
        return torch.empty((B, H, W, 3), dtype=torch.uint8, device=device)
    if out == "f16":
        return torch.empty((B, 3, H, W), dtype=torch.float16, device=device)
    if out == "f32":
        return torch.empty((B, 3, H, W), dtype=torch.float32, device=device)
    raise ValueError("out must be 'u8', 'f16' or 'f32'")


def lanczos_plan_host(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow's coefficient table for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc), computed by libgswm on the host."""
    lib = N.lib()
    ksize = lib.gsw_lanczos_plan(in_size, out_size, None, None, 0)
    if ksize <= 0:
        N.check(-ksize)
    bounds = np.empty((out_size, 2), dtype=np.int32)
    kk = np.empty((out_size, ksize), dtype=np.int32)
    rc = lib.gsw_lanczos_plan(in_size, out_size, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), kk.size)
    if rc <= 0:
        N.check(-rc)
    return bounds, kk, ksize


def _plan(in_size: int, out_size: int, device) -> Tuple[torch.Tensor, torch.Tensor, int]:
    k = (in_size, out_size, str(device))
    if k not in _PLANS:
        bounds, kk, ksize = lanczos_plan_host(in_size, out_size)
        _PLANS[k] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    return _PLANS[k]


def resize_lanczos(images: torch.Tensor, size: Union[int, Tuple[int, int], None], *, out: str = "u8") -> torch.Tensor:
    """`pil_img.resize(size, Image.Resampling.LANCZOS)` for a batch (size = (width, height) as PIL takes it; None keeps the size),
    with the output conversion fused: out = 'u8' (PIL image), 'f32' (ToTensor) or 'f16' (the reference's VAE-encoder input)."""
    images = _check_images(images)
    B, H, W, _ = images.shape
    if size is None:
        size = (W, H)
    if isinstance(size, int):
        size = (size, size)
    Wo, Ho = int(size[0]), int(size[1])
    dev = images.device
    res = _alloc_out(B, Ho, Wo, out, dev)
    hb = hk = vb = vk = None
    hks = vks = 0
    if Wo != W:
        hb, hk, hks = _plan(W, Wo, dev)
    if Ho != H:
        vb, vk, vks = _plan(H, Ho, dev)
    tmp = torch.empty((B, H, Wo, 3), dtype=torch.uint8, device=dev) if (Wo != W and (Ho != H or out != "u8")) else None
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_resize_lanczos(images.data_ptr(), B, H, W, res.data_ptr(), Ho, Wo, _MODES[out], tmp.data_ptr() if tmp is not None else None,
                                           hb.data_ptr() if hb is not None else None, hk.data_ptr() if hk is not None else None, hks,
                                           vb.data_ptr() if vb is not None else None, vk.data_ptr() if vk is not None else None, vks, _stream_ptr()))
    return res


def to_tensor(images: torch.Tensor, *, out: str = "f16") -> torch.Tensor:
    """ToTensor (+ fp16 cast and 2x-1 when out='f16') without resizing."""
    return resize_lanczos(images, None, out=out)


def tensor_to_image(x: torch.Tensor, *, denormalise: bool = False) -> torch.Tensor:
    """[B, 3, H, W] float tensor -> uint8 [B, H, W, 3] like numpy_to_pil: (x * 255).round(); denormalise=True first applies the
    pipeline's (x / 2 + 0.5).clamp(0, 1)."""
    if x.dim() != 4 or x.shape[1] != 3 or not x.is_cuda:
        raise ValueError("x must be a [B, 3, H, W] device tensor")
    x = x.contiguous()
    B, _, H, W = x.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_tensor_to_image(x.data_ptr(), _dt(x.dtype), B, H, W, 1 if denormalise else 0, out.data_ptr(), _stream_ptr()))
    return out


def jpeg_quant_tables(quality: int) -> Tuple[np.ndarray, np.ndarray]:
    lum = np.empty(64, dtype=np.uint8)
    chrom = np.empty(64, dtype=np.uint8)
    N.check(N.lib().gsw_jpeg_quant_tables(int(quality), lum.ctypes.data_as(C.c_void_p), chrom.ctypes.data_as(C.c_void_p)))
    return lum.reshape(8, 8), chrom.reshape(8, 8)


def jpeg_roundtrip(images: torch.Tensor, quality: int, *, out: str = "u8") -> torch.Tensor:
    """distortions:175-184: what `Image.open(BytesIO(image.save(format="JPEG", quality=q)))` decodes to, for a batch, on the device."""
    images = _check_images(images)
    B, H, W, _ = images.shape
    dev = images.device
    res = _alloc_out(B, H, W, out, dev)
    ws = torch.empty(N.lib().gsw_jpeg_workspace_bytes(B, H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_jpeg_roundtrip(images.data_ptr(), B, H, W, int(quality), res.data_ptr(), _MODES[out], ws.data_ptr(), _stream_ptr()))
    return res


def gaussian_blur_params(radius: float, passes: int = 3) -> Tuple[int, int, int]:
    """(box radius, ww, fw) Pillow derives from a Gaussian radius (BoxBlur.c), computed by libgswm on the host."""
    r, ww, fw = C.c_int(), C.c_uint32(), C.c_uint32()
    N.check(N.lib().gsw_gaussian_blur_params(float(radius), int(passes), C.byref(r), C.byref(ww), C.byref(fw)))
    return r.value, ww.value, fw.value


def gaussian_blur(images: torch.Tensor, radius: float) -> torch.Tensor:
    """distortions:157-164: `image.filter(ImageFilter.GaussianBlur(radius))` for a batch, bit-exact, on the device."""
    images = _check_images(images)
    B, H, W, _ = images.shape
    out, tmp = torch.empty_like(images), torch.empty_like(images)
    with torch.cuda.device(images.device):
        N.check(N.lib().gsw_gaussian_blur(images.data_ptr(), B, H, W, float(radius), out.data_ptr(), tmp.data_ptr(), _stream_ptr()))
    return out


_OPS = {"brightness": N.GSW_PW_BRIGHTNESS, "contrast": N.GSW_PW_CONTRAST, "invert": N.GSW_PW_INVERT, "togray": N.GSW_PW_GRAY,
        "horizontal_flip": N.GSW_PW_HFLIP, "vertical_flip": N.GSW_PW_VFLIP, "noise": N.GSW_PW_NOISE}


def pointwise(images: torch.Tensor, op: str, strength: float = 0.0, *, seed: int = 0, image_index0: int = 0, out: str = "u8") -> torch.Tensor:
    images = _check_images(images)
    B, H, W, _ = images.shape
    dev = images.device
    res = _alloc_out(B, H, W, out, dev)
    ws = torch.empty(B, dtype=torch.int64, device=dev) if op == "contrast" else None
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_image_pointwise(images.data_ptr(), B, H, W, _OPS[op], float(strength), int(seed), int(image_index0), res.data_ptr(),
                                            _MODES[out], ws.data_ptr() if ws is not None else None, _stream_ptr()))
    return res


# the strength ranges of the reference's tool (distortions:17-34), for the distortion types this module runs on the device
distortion_strength_paras = dict(scaling=(0, 1), brightness=(1, 16), contrast=(1, 6), blurring=(0, 20), noise=(0, 0.5), compression=(100, 0),
                                 horizontal_flip=(0, 0), vertical_flip=(0, 0), togray=(0, 0), invert=(0, 0))


def relative_strength_to_absolute(strength: float, distortion_type: str) -> float:
    """distortions:37-49."""
    assert 0 <= strength <= 1
    lo, hi = distortion_strength_paras[distortion_type]
    s = strength * (hi - lo) + lo
    s = max(s, min(lo, hi))
    return min(s, max(lo, hi))


def apply_distortion(images: torch.Tensor, distortion_type: str, strength: Optional[float] = None, *, distortion_seed: int = 0,
                     relative_strength: bool = True, out: str = "u8") -> torch.Tensor:
    """Batch form of distortions:52-233 `apply_distortion` for the device-resident types; `strength` follows the reference
    (relative in [0, 1] unless relative_strength=False).  Unlike the reference the whole batch is one launch; for "noise" the image
    index keys the generator (the reference increments the seed per image)."""
    if distortion_type not in distortion_strength_paras:
        raise ValueError(f"distortion type {distortion_type!r} is not implemented on the device")
    if strength is not None and relative_strength:
        strength = relative_strength_to_absolute(strength, distortion_type)
    lo, hi = distortion_strength_paras[distortion_type]
    if strength is not None:
        assert min(lo, hi) <= strength <= max(lo, hi)
    if distortion_type == "compression":
        return jpeg_roundtrip(images, int(strength), out=out)
    if distortion_type == "blurring":
        blurred = gaussian_blur(images, int(strength))                  # distortions:158-164: kernel_size = int(strength)
        return blurred if out == "u8" else to_tensor(blurred, out=out)
    if distortion_type == "scaling":
        _, H, W, _ = images.shape
        return resize_lanczos(images, (int(W * strength), int(H * strength)), out=out)
    return pointwise(images, distortion_type, 0.0 if strength is None else strength, seed=distortion_seed, out=out)
