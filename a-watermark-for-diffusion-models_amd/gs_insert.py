"""Drop-in twin of the reference's gs_insert.py: same function name, arguments, return type and side effect,
with the per-element hot loop (gs_insert.py:49-66) running as one HIP kernel on the MI355X.

    from gswm_amd.gs_insert import gs_watermark_init_noise
    Z_s_T_arrays = [gs_watermark_init_noise(opt, opt.message) for _ in range(opt.n_samples)]     # README.md:110-112
"""
from __future__ import annotations

from datetime import datetime

import numpy as np
import torch

from . import codec


def _write_info(path, key: bytes, nonce: bytes, k: bytes, extra=()):
    # gs_insert.py:68-74 (extra: nodes.py:130-135)
    with open(path, "a") as f:
        f.write(f"Time: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
        f.write(f"key: {key.hex()}\n")
        f.write(f"nonce: {nonce.hex()}\n")
        f.write(f"message: {k.hex()}\n")
        for line in extra:
            f.write(line + "\n")
        f.write("----------------------\n")


def gs_watermark_init_noise(opt, message="", *, log_path="info_data.txt", device="cuda"):
    """gs_insert.py:8-75.  `opt` is any object with str attributes key_hex / nonce_hex.

    Returns a float64 ndarray (4, 64, 64).  The uniforms are drawn from the GLOBAL numpy RNG exactly like the reference
    (`np.random.uniform(0, 1)` per element == `np.random.uniform(0, 1, N)`), so seeding numpy reproduces the reference's
    output (and leaves the global generator where the reference leaves it); the MT19937 draws, ChaCha20, bit expansion and
    norm.ppf all run on the GPU (Cephes ndtri in fp64).
    """
    k = codec.pad_message(message, 32)                                  # :9-20
    key, nonce = codec.resolve_key_nonce(opt.key_hex, opt.nonce_hex)    # :27-42
    u_dev = codec.mt19937_uniform(4 * 64 * 64, device=device)           # :62: the 16384 np.random.uniform(0, 1) draws, made on the device
    z = codec.embed_batch(key, nonce, k, 1, (4, 64, 64), u=u_dev.view(1, -1), dtype=torch.float64, device=device)
    out = z[0].cpu().numpy()
    if log_path:
        _write_info(log_path, key, nonce, k)                            # :68-74
    return out


def gs_watermark_init_noise_batch(opt, message="", n_samples=1, *, dtype=torch.float32, device="cuda", seed=None,
                                  image_index0=0, fast=False, log_path=None):
    """Batch form of the README call site (`torch.stack([torch.tensor(a).float() ...]).to(device)`, README.md:110-112):
    returns the stacked [n_samples, 4, 64, 64] device tensor directly.

    seed=None: uniforms come from the global numpy RNG (bit-parity with n_samples reference calls in a row);
    seed=int: in-kernel Philox stream, nothing crosses PCIe.
    """
    k = codec.pad_message(message, 32)
    key, nonce = codec.resolve_key_nonce(opt.key_hex, opt.nonce_hex)
    u_dev = None
    if seed is None:
        u_dev = codec.mt19937_uniform(n_samples * 16384, device=device).view(n_samples, -1)
    z = codec.embed_batch(key, nonce, k, n_samples, (4, 64, 64), u=u_dev, seed=seed or 0, image_index0=image_index0,
                          dtype=dtype, fast=fast, device=device)
    if log_path:
        for _ in range(n_samples):
            _write_info(log_path, key, nonce, k)
    return z
