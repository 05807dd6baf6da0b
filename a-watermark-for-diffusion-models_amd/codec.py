"""Batch-first device API of the Gaussian-Shading codec: thin, stream-ordered wrappers over the C ABI.

Everything here runs on the current HIP device through libgswm.so; tensors are only used for device memory and
stream plumbing (`data_ptr()`, `torch.cuda.current_stream()`).  There is no CPU path.

Reference rows: E1-E6 -> `embed_batch`, X3-X5 -> `extract_batch`, X6 -> `bit_matches`,
X2/G1 elementwise step -> `ddim_step*`.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N

_DTYPES = {torch.float32: N.GSW_F32, torch.float16: N.GSW_F16, torch.bfloat16: N.GSW_BF16, torch.float64: N.GSW_F64}


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); the watermark hot path has no CPU fallback")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if t.data_ptr() % 16:
        raise ValueError(f"{name} must be 16-byte aligned (got a view at an odd storage offset; .clone() it)")


def _like(t: torch.Tensor, ref: torch.Tensor, name: str, numel: Optional[int] = None):
    """A companion operand whose raw pointer crosses the C ABI next to `ref`: same device and dtype, contiguous, aligned, expected size.
    (A dtype mismatch would be read as the wrong bytes, a short tensor is an out-of-bounds device read: both are silent otherwise.)"""
    _need_gpu(t, name)
    if t.device != ref.device:
        raise RuntimeError(f"{name} lives on {t.device}, expected {ref.device}")
    if t.dtype != ref.dtype:
        raise ValueError(f"{name} is {t.dtype}, expected {ref.dtype}")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name} has {t.numel()} elements, expected {numel}")


def _dt(t: torch.dtype) -> int:
    try:
        return _DTYPES[t]
    except KeyError:
        raise ValueError(f"unsupported dtype {t}") from None


def _check_key_nonce(key: bytes, nonce: bytes):
    # `cryptography` raises ValueError for wrong sizes (gs_insert.py:45)
    if len(key) != 32:
        raise ValueError("ChaCha20 key must be 32 bytes (256 bits)")
    if len(nonce) != 16:
        raise ValueError("ChaCha20 nonce must be 16 bytes (128 bits)")


# ------------------------------------------------------------------------------------------------ E1: host-side prep
def pad_message(message: str, msg_bytes: int = 32) -> bytes:
    """gs_insert.py:9-20 / nodes.py:68-76: UTF-8, zero-pad or truncate to msg_bytes; empty -> os.urandom."""
    if message:
        b = str(message).encode()
        return b + b"\x00" * (msg_bytes - len(b)) if len(b) < msg_bytes else b[:msg_bytes]
    return os.urandom(msg_bytes)


def resolve_key_nonce(key_hex: str, nonce_hex: str) -> Tuple[bytes, bytes]:
    """gs_insert.py:27-42: both given; key only -> nonce = key bytes 8..23; neither -> random."""
    if key_hex and nonce_hex:
        return bytes.fromhex(key_hex), bytes.fromhex(nonce_hex)
    if key_hex and not nonce_hex:
        return bytes.fromhex(key_hex), bytes.fromhex(key_hex[16:48])
    return os.urandom(32), os.urandom(16)


def choose_watermark_length(total_blocks_needed: int) -> int:
    """nodes.py:26-49."""
    for bits in (1024, 512, 256, 128, 64):
        if total_blocks_needed >= bits * 32:
            return bits
    return 32


# ------------------------------------------------------------------------------------------------ E2
def keystream(key: bytes, nonce: bytes, nbytes: int, device="cuda") -> torch.Tensor:
    """ChaCha20 keystream bytes (OpenSSL 16-byte nonce layout) as a uint8 device tensor."""
    _check_key_nonce(key, nonce)
    out = torch.empty(nbytes, dtype=torch.uint8, device=device)
    with torch.cuda.device(out.device):
        N.check(N.lib().gsw_keystream(key, nonce, out.data_ptr(), nbytes, _stream_ptr()))
    return out


# ------------------------------------------------------------------------------------------------ E3-E6
def embed_batch(key: bytes, nonce: bytes, k: bytes, batch: int, shape: Sequence[int], *, u: Optional[torch.Tensor] = None,
                seed: int = 0, image_index0: int = 0, dtype: torch.dtype = torch.float32, fast: bool = False,
                device="cuda", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Watermarked initial latents Z_s_T, shape [batch, *shape] (shape = (4, H/8, W/8)).

    u: optional float64 device tensor [batch, prod(shape)] of uniforms (the reference's np.random.uniform draws) for
       bit-parity with gs_insert.py:62-64; None -> in-kernel Philox4x32-7 keyed by (seed, image_index0 + b, element).
    fast: fp32 inverse-CDF core (|dz| <= 1e-5) instead of Cephes fp64.
    """
    _check_key_nonce(key, nonce)
    n = 1
    for s in shape:
        n *= int(s)
    if out is None:
        out = torch.empty((batch, *shape), dtype=dtype, device=device)
    else:
        if out.numel() != batch * n:
            raise ValueError("out has the wrong size")
    _need_gpu(out, "out")
    u_ptr = None
    if u is not None:
        _need_gpu(u, "u")
        if u.dtype != torch.float64 or u.numel() != batch * n:
            raise ValueError("u must be float64 with batch*n_elems entries")
        u_ptr = u.data_ptr()
    with torch.cuda.device(out.device):
        N.check(N.lib().gsw_embed(key, nonce, k, len(k), u_ptr, seed & (2**64 - 1), image_index0, out.data_ptr(), _dt(out.dtype),
                                  batch, n, N.GSW_EMBED_FAST_F32 if fast else N.GSW_EMBED_EXACT_F64, _stream_ptr()))
    return out


def philox_uniform(seed: int, image_index0: int, batch: int, n_elems: int, device="cuda") -> torch.Tensor:
    """The u stream the embed kernel draws when no `u` is supplied ([batch, n_elems] float64)."""
    out = torch.empty((batch, n_elems), dtype=torch.float64, device=device)
    with torch.cuda.device(out.device):
        N.check(N.lib().gsw_philox_uniform(seed & (2**64 - 1), image_index0, out.data_ptr(), batch, n_elems, _stream_ptr()))
    return out


def mt19937_seed(seed: int) -> np.ndarray:
    """NumPy's legacy integer seeding (RandomState(seed) / np.random.seed(seed)): the 624 key words; pos starts at 624."""
    if not 0 <= int(seed) <= 0xFFFFFFFF:
        raise ValueError("Seed must be between 0 and 2**32 - 1")
    key = np.empty(624, dtype=np.uint32)
    N.lib().gsw_mt19937_seed(int(seed), key.ctypes.data)
    return key


def mt19937_uniform(n: int, rng=None, *, device="cuda") -> torch.Tensor:
    """n draws of `rng.uniform(0, 1)` (== legacy `random_sample`) generated ON THE DEVICE from the generator's current state:
    returns float64 [n] on the device and advances `rng` (a np.random.RandomState, or None for NumPy's global generator) exactly as
    the n host draws would -- gs_insert.py:62 / nodes.py:114-117 without shipping the uniforms over PCIe."""
    target = np.random if rng is None else rng
    state = target.get_state()
    if state[0] != "MT19937":
        raise ValueError("only the legacy MT19937 generator is supported")
    key = np.ascontiguousarray(state[1], dtype=np.uint32)
    out = torch.empty(int(n), dtype=torch.float64, device=device)
    st = torch.empty(625, dtype=torch.int32, device=device)
    with torch.cuda.device(out.device):
        N.check(N.lib().gsw_mt19937_uniform(key.ctypes.data, int(state[2]), out.data_ptr(), int(n), st.data_ptr(), _stream_ptr()))
    new = st.cpu().numpy().view(np.uint32)
    target.set_state(("MT19937", new[:624].copy(), int(new[624]), state[3], state[4]))
    return out


# ------------------------------------------------------------------------------------------------ X3-X5
def extract_batch(z: torch.Tensor, key: bytes, nonce: bytes, message_length: int, *, return_counts: bool = False):
    """Recover the message from latents z [B, ...] (any of fp16/bf16/fp32/fp64).

    Returns (bits uint8 [B, ceil(M/8)] MSB-first, flags int32 [B]) (+ counts int32 [B, M] '1'-votes).
    flags != 0 marks images for which the reference raises ValueError (saturated cdf / NaN), extract.py:84-86.
    Raises IndexError when the reference would (padded bit count not a multiple of message_length).
    """
    _check_key_nonce(key, nonce)
    _need_gpu(z, "z")
    B = z.shape[0]
    n = z.numel() // max(B, 1)
    M = int(message_length)
    bits = torch.empty((B, (M + 7) // 8), dtype=torch.uint8, device=z.device)
    flags = torch.empty((B,), dtype=torch.int32, device=z.device)
    counts = torch.empty((B, M), dtype=torch.int32, device=z.device) if return_counts else None
    with torch.cuda.device(z.device):
        N.check(N.lib().gsw_extract(z.data_ptr(), _dt(z.dtype), key, nonce, M, bits.data_ptr(),
                                    counts.data_ptr() if return_counts else None, flags.data_ptr(), B, n, _stream_ptr()))
    return (bits, flags, counts) if return_counts else (bits, flags)


def bit_matches(bits: torch.Tensor, message_length: int, ref_msg: bytes, ref_bits: Optional[int] = None) -> torch.Tensor:
    """Per-image count of bits equal to ref_msg over min(message_length, ref_bits) positions (extract.py:103-110)."""
    _need_gpu(bits, "bits")
    B = bits.shape[0]
    out = torch.empty((B,), dtype=torch.int32, device=bits.device)
    rb = 8 * len(ref_msg) if ref_bits is None else ref_bits
    with torch.cuda.device(bits.device):
        N.check(N.lib().gsw_bit_matches(bits.data_ptr(), int(message_length), ref_msg, rb, out.data_ptr(), B, _stream_ptr()))
    return out


def bits_to_str(bits_row) -> str:
    """uint8 bytes (MSB-first) -> '0'/'1' string, the reference's return type (extract.py:101)."""
    return "".join(format(int(b), "08b") for b in bits_row)


# ------------------------------------------------------------------------------------------------ X2 / G1 elementwise
def ddim_step(x: torch.Tensor, model_out: torch.Tensor, a: float, b: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a*x + b*model_out (fp32 math, one rounding). out may be x (in place)."""
    _need_gpu(x, "x"); _need_gpu(model_out, "model_out")
    if model_out.dtype != x.dtype or model_out.numel() != x.numel():
        raise ValueError("x / model_out mismatch")
    if out is None:
        out = torch.empty_like(x)
    _like(out, x, "out", x.numel())
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_ddim_step(x.data_ptr(), model_out.data_ptr(), out.data_ptr(), a, b, _dt(x.dtype), x.numel(), _stream_ptr()))
    return out


def ddim_step_cfg(x: torch.Tensor, e_uncond: torch.Tensor, e_text: torch.Tensor, a: float, b: float, guidance: float,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = a*x + b*(e_uncond + guidance*(e_text - e_uncond))."""
    _need_gpu(x, "x")
    _like(e_uncond, x, "e_uncond", x.numel()); _like(e_text, x, "e_text", x.numel())
    if out is None:
        out = torch.empty_like(x)
    _like(out, x, "out", x.numel())
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_ddim_step_cfg(x.data_ptr(), e_uncond.data_ptr(), e_text.data_ptr(), out.data_ptr(), a, b, guidance,
                                          _dt(x.dtype), x.numel(), _stream_ptr()))
    return out


def ddim_step_extract(x: torch.Tensor, model_out: torch.Tensor, a: float, b: float, key: bytes, nonce: bytes,
                      message_length: int, *, z_out: Optional[torch.Tensor] = None, return_counts: bool = False):
    """Last inversion step fused with the vote: z = a*x + b*model_out is quantised and voted without a round trip to HBM."""
    _check_key_nonce(key, nonce)
    _need_gpu(x, "x")
    _like(model_out, x, "model_out", x.numel())
    if z_out is not None:
        _like(z_out, x, "z_out", x.numel())
    B = x.shape[0]
    n = x.numel() // max(B, 1)
    M = int(message_length)
    bits = torch.empty((B, (M + 7) // 8), dtype=torch.uint8, device=x.device)
    flags = torch.empty((B,), dtype=torch.int32, device=x.device)
    counts = torch.empty((B, M), dtype=torch.int32, device=x.device) if return_counts else None
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_ddim_step_extract(x.data_ptr(), model_out.data_ptr(), z_out.data_ptr() if z_out is not None else None,
                                              a, b, _dt(x.dtype), key, nonce, M, bits.data_ptr(),
                                              counts.data_ptr() if return_counts else None, flags.data_ptr(), B, n, _stream_ptr()))
    return (bits, flags, counts) if return_counts else (bits, flags)


# ------------------------------------------------------------------------------------------------ eps-model fusions (X2 / G1)
def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float, *, act: bool = True,
                   pre_bias: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = act(GroupNorm(x + pre_bias[:, :, None, None]) * gamma + beta) for NCHW x in one kernel (2 HBM passes)."""
    _need_gpu(x, "x")
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // max(B * C, 1)
    if out is None:
        out = torch.empty_like(x)
    _like(out, x, "out", x.numel())
    _like(gamma, x, "gamma", C); _like(beta, x, "beta", C)
    pb = None
    if pre_bias is not None:
        pre_bias = pre_bias.to(x.dtype).contiguous()
        _like(pre_bias, x, "pre_bias", B * C)
        pb = pre_bias.data_ptr()
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_groupnorm_silu(x.data_ptr(), pb, gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), B, C, HW, groups, eps,
                                           1 if act else 0, _dt(x.dtype), _stream_ptr()))
    return out


def geglu(x: torch.Tensor) -> torch.Tensor:
    """[..., 2*I] -> [..., I]: x[..., :I] * gelu(x[..., I:]) in one pass."""
    _need_gpu(x, "x")
    inner = x.shape[-1] // 2
    rows = x.numel() // (2 * inner)
    out = torch.empty((*x.shape[:-1], inner), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_geglu(x.data_ptr(), out.data_ptr(), rows, inner, _dt(x.dtype), _stream_ptr()))
    return out


def add_layernorm(x: torch.Tensor, delta: Optional[torch.Tensor], weight: torch.Tensor, bias: torch.Tensor, eps: float):
    """(x + delta, LayerNorm(x + delta)) in one kernel; delta=None -> (x, LayerNorm(x))."""
    _need_gpu(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    _like(weight, x, "weight", C); _like(bias, x, "bias", C)
    y = torch.empty_like(x)
    xnew = x
    dptr = None
    if delta is not None:
        _like(delta, x, "delta", x.numel())
        xnew = torch.empty_like(x)
        dptr = delta.data_ptr()
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_add_layernorm(x.data_ptr(), dptr, weight.data_ptr(), bias.data_ptr(), xnew.data_ptr() if delta is not None else None,
                                          y.data_ptr(), rows, C, eps, _dt(x.dtype), _stream_ptr()))
    return xnew, y
