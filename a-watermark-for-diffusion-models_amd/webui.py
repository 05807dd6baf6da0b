"""Adapter for the A1111 WebUI scripts of the reference (scripts/GS_watermark_insert_for_webui_v1.6.0_and_higher.py:26-190): the object the
script swaps in for `modules.rng.ImageRNG`, and the codec options of its `init_gs_Z_s_T` (32-byte message, or 8 bytes repeated four times with
`use_repeat`; RandomState(randomSeed) or the global numpy stream).  Nothing here imports the WebUI: `install()` takes the `modules.rng` module.
The watermarked first noise comes from ONE device launch for the whole batch (the reference returns a single (1, 4, 64, 64) lattice for any batch)."""
from __future__ import annotations

import os

import numpy as np
import torch

from datetime import datetime

from . import codec


def _write_info(path, key: bytes, nonce: bytes, k: bytes, randomSeed):
    # scripts/...v1.6.0...py:82-89: the seed line sits between nonce and message
    with open(path, "a") as f:
        f.write(f"Time: {datetime.now().strftime('%Y-%m-%d %H:%M:%S')}\n")
        f.write(f"key: {key.hex()}\nnonce: {nonce.hex()}\nrandomSeed: {randomSeed}\nmessage: {k.hex()}\n")
        f.write("----------------------\n")


def init_gs_Z_s_T(message="", key_hex="", nonce_hex="", *, use_randomSeed=0, randomSeed=42, use_repeat=0, batch=1, shape=(4, 64, 64),
                  device="cuda", log_path="info_data.txt") -> torch.Tensor:
    """scripts/...v1.6.0...py:26-91 for `batch` images -> float32 [batch, 4, 64, 64] on `device`.  An empty message draws random bytes, an empty key a
    random key / nonce (as the script does)."""
    nbytes = 8 if int(use_repeat) == 1 else 32
    k = codec.pad_message(message, nbytes) if message else os.urandom(nbytes)
    if int(use_repeat) == 1:
        k = k * 4
    if key_hex:
        key, nonce = codec.resolve_key_nonce(key_hex, nonce_hex)
    else:
        key, nonce = os.urandom(32), os.urandom(16)
    n = int(np.prod(shape))
    rng = np.random.RandomState(seed=int(randomSeed)) if int(use_randomSeed) != 0 else None
    u = codec.mt19937_uniform(n * batch, rng, device=device).view(batch, -1)
    z = codec.embed_batch(key, nonce, k, batch, tuple(shape), u=u, dtype=torch.float32, device=device)
    if log_path:
        _write_info(log_path, key, nonce, k, randomSeed)
    return z


class GaussianShadingImageRNG:
    """Same constructor and `first()` / `next()` protocol as the WebUI's `modules.rng.ImageRNG` (scripts/...py:118-147): the first noise of a job is the
    watermarked latent, later calls fall back to per-seed torch generators."""

    options = dict(message="", key_hex="", nonce_hex="", use_randomSeed=0, randomSeed=42, use_repeat=0)

    def __init__(self, shape, seeds, subseeds=None, subseed_strength=0.0, seed_resize_from_h=0, seed_resize_from_w=0, *, device="cuda"):
        self.shape = tuple(map(int, shape))
        self.seeds, self.subseeds, self.subseed_strength = seeds, subseeds, subseed_strength
        self.seed_resize_from_h, self.seed_resize_from_w = seed_resize_from_h, seed_resize_from_w
        self.device = device
        self.generators = [torch.Generator("cpu").manual_seed(int(s)) for s in seeds]
        self.is_first = True

    def first(self):
        return init_gs_Z_s_T(batch=len(self.seeds), shape=self.shape, device=self.device, **self.options)

    def next(self):
        if self.is_first:
            self.is_first = False
            return self.first()
        return torch.stack([torch.randn(self.shape, generator=g) for g in self.generators]).to(self.device)


def install(rng_module, **options):
    """`rng_module.ImageRNG = GaussianShadingImageRNG` with the script's options (message, key_hex, nonce_hex, use_randomSeed, randomSeed, use_repeat);
    returns the previous class so the caller can restore it."""
    unknown = set(options) - set(GaussianShadingImageRNG.options)
    if unknown:
        raise TypeError(f"unknown options {sorted(unknown)}")
    GaussianShadingImageRNG.options = {**GaussianShadingImageRNG.options, **options}
    previous = rng_module.ImageRNG
    rng_module.ImageRNG = GaussianShadingImageRNG
    return previous
