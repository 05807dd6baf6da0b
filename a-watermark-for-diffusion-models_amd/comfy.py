"""Twin of the codec function in the reference's ComfyUI_GSWaterMark/nodes.py (the generalised H x W lattice).
Only the codec is mirrored; the ComfyUI node classes are host-UI glue and out of scope."""
from __future__ import annotations

import numpy as np
import torch

from . import codec
from .gs_insert import _write_info

choose_watermark_length = codec.choose_watermark_length  # nodes.py:26-49


def gs_watermark_init_noise(key_hex, nonce_hex, device, message, use_seed, randomSeed, width, height, message_length=-1,
                            *, log_path="info_data.txt", compute_device="cuda"):
    """nodes.py:51-138.  Returns a float32 CPU tensor (4, height//8, width//8) like the reference.

    use_seed == 1 -> RandomState(randomSeed) (nodes.py:52-53), else the global numpy stream (nodes.py:114-115).
    """
    h, w = height // 8, width // 8
    n = 4 * h * w
    bits = message_length if message_length != -1 else choose_watermark_length(n)      # :61-66
    k = codec.pad_message(message, bits // 8)                                           # :68-76
    key, nonce = codec.resolve_key_nonce(key_hex, nonce_hex)                            # :90-99
    rng = np.random.RandomState(seed=randomSeed) if int(use_seed) == 1 else None
    u = codec.mt19937_uniform(n, rng, device=compute_device).view(1, -1)           # :114-117, drawn on the device
    z = codec.embed_batch(key, nonce, k, 1, (4, h, w), u=u, dtype=torch.float32, device=compute_device)
    if log_path:
        _write_info(log_path, key, nonce, k, extra=(f"randomSeed: {randomSeed}", f"height: {height}", f"width: {width}",
                                                    f"randomSeed: {randomSeed}", f"message_length: {message_length}"))
    return z[0].cpu()
