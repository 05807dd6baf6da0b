"""Drop-in twin of the recover-bits API of the reference's extract.py (extract.py:72-110), with the per-element
norm.cdf loop, the ChaCha20 decrypt and the majority vote running as one HIP kernel.

`args` is the reference's argparse namespace: .key (32 bytes), .nonce (16 bytes), .l (must be 1: the reference's
l > 1 path is non-functional), .message_length.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N
from . import codec

_TORCH_OK = (torch.float16, torch.bfloat16, torch.float32, torch.float64)


def _to_device_latents(reversed_latents, device):
    """np.nditer order (extract.py:82): memory order of the array, i.e. C order for the contiguous tensors the
    inversion returns.  One image per call, any shape."""
    if isinstance(reversed_latents, torch.Tensor):
        t = reversed_latents.detach()
        if t.dtype not in _TORCH_OK:
            t = t.to(torch.float64)
        return t.contiguous().to(device).reshape(1, -1)
    a = np.asarray(reversed_latents)
    if a.dtype not in (np.float16, np.float32, np.float64):
        a = a.astype(np.float64)
    a = np.ascontiguousarray(a.ravel(order="K"))
    return torch.from_numpy(a).to(device).reshape(1, -1)


def recover_exactracted_message(reversed_latents, args, *, device="cuda"):
    """extract.py:72-101 -> str of message_length '0'/'1' characters.

    Raises ValueError where the reference does (a latent >= 8.2924 saturates norm.cdf so int(y) == 2, or NaN;
    extract.py:84-86) and IndexError when the padded bit count is not a multiple of message_length (extract.py:98).
    """
    if int(getattr(args, "l", 1)) != 1:
        raise ValueError("only l == 1 is functional in the reference (extract.py:84-86 breaks for l > 1)")
    z = _to_device_latents(reversed_latents, device)
    m = int(args.message_length)
    bits, flags = codec.extract_batch(z, args.key, args.nonce, m)
    f = int(flags[0].item())
    if f & N.GSW_FLAG_NAN:
        raise ValueError("cannot convert float NaN to integer")
    if f & N.GSW_FLAG_SATURATED:
        raise ValueError("invalid literal for int() with base 2")
    return codec.bits_to_str(bits[0].cpu().numpy())[:m]


def recover_exactracted_message_batch(latents: torch.Tensor, args):
    """Batch form: latents [B, 4, h, w] on the device -> (list of bit strings or the raised exception per image).
    Mirrors the per-image try/except of extract.py:148-155."""
    m = int(args.message_length)
    bits, flags = codec.extract_batch(latents.contiguous(), args.key, args.nonce, m)
    bits_h, flags_h = bits.cpu().numpy(), flags.cpu().numpy()
    out = []
    for b in range(bits_h.shape[0]):
        if flags_h[b] & N.GSW_FLAG_NAN:
            out.append(ValueError("cannot convert float NaN to integer"))
        elif flags_h[b] & N.GSW_FLAG_SATURATED:
            out.append(ValueError("invalid literal for int() with base 2"))
        else:
            out.append(codec.bits_to_str(bits_h[b])[:m])
    return out


def calculate_bit_accuracy(original_message_hex, extracted_message_bin):
    """extract.py:103-110 (host string arithmetic; the batched device form is codec.bit_matches)."""
    original_message_bin = bin(int(original_message_hex, 16))[2:].zfill(len(original_message_hex) * 4)
    min_length = min(len(original_message_bin), len(extracted_message_bin))
    original_message_bin = original_message_bin[:min_length]
    extracted_message_bin = extracted_message_bin[:min_length]
    matching_bits = sum(1 for x, y in zip(original_message_bin, extracted_message_bin) if x == y)
    return original_message_bin, matching_bits / min_length
