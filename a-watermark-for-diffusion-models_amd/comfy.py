"""Twin of the reference's ComfyUI_GSWaterMark/nodes.py: the codec on the generalised H x W lattice (nodes.py:26-138) and thin node classes with the
reference's names, inputs, outputs and categories (nodes.py:167-252) so an existing workflow JSON keeps loading.  The node bodies are this build's:
the whole batch is embedded by one device launch; ComfyUI itself is only imported inside GSKSamplerAdvanced.sample."""
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


MAX_RESOLUTION = 16384          # ComfyUI's nodes.MAX_RESOLUTION


class GSLatent:
    """nodes.py:209-240 "GS Latent Noise": watermarked initial latents for a batch.  use_seed == 1 -> every image of the batch is the SAME lattice
    (the reference re-seeds RandomState per call, nodes.py:232-235); otherwise the global numpy stream advances by N draws per image."""

    @classmethod
    def INPUT_TYPES(cls):
        return {"required": {
            "use_seed": ("INT", {"default": 1, "min": 0, "max": 1}),
            "seed": ("INT", {"default": 42, "min": 0, "max": 0xffffffff}),
            "width": ("INT", {"default": 512, "min": 64, "max": MAX_RESOLUTION, "step": 8}),
            "height": ("INT", {"default": 512, "min": 64, "max": MAX_RESOLUTION, "step": 8}),
            "key": ("STRING", {"default": "5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7"}),
            "nonce": ("STRING", {"default": "05072fd1c2265f6f2e2a4080a2bfbdd8"}),
            "message": ("STRING", {"default": "lthero"}),
            "message_length": ("INT", {"default": -1, "min": 32, "max": 1024, "step": 32}),
            "batch_size": ("INT", {"default": 1, "min": 1, "max": 64}),
        }}

    RETURN_TYPES = ("LATENT", "IMAGE")
    FUNCTION = "create_gs_latents"
    CATEGORY = "GSWatermark-lthero/latent/noise"

    def create_gs_latents(self, key, nonce, message, batch_size, use_seed, seed, width, height, message_length, *, log_path="info_data.txt",
                          compute_device="cuda"):
        h, w = height // 8, width // 8
        n = 4 * h * w
        bits = message_length if message_length != -1 else choose_watermark_length(n)
        k = codec.pad_message(message, bits // 8)
        kb, nb = codec.resolve_key_nonce(key, nonce)
        B = int(batch_size)
        if int(use_seed) == 1:
            u = codec.mt19937_uniform(n, np.random.RandomState(seed=seed), device=compute_device).view(1, -1)
            z = codec.embed_batch(kb, nb, k, 1, (4, h, w), u=u, dtype=torch.float32, device=compute_device).expand(B, -1, -1, -1)
        else:
            u = codec.mt19937_uniform(n * B, None, device=compute_device).view(B, -1)            # image b takes draws [b n, (b+1) n) of the global stream
            z = codec.embed_batch(kb, nb, k, B, (4, h, w), u=u, dtype=torch.float32, device=compute_device)
        if log_path:
            for _ in range(1 if int(use_seed) == 1 else B):                                      # one log block per codec call of the reference
                _write_info(log_path, kb, nb, k, extra=(f"randomSeed: {seed}", f"height: {height}", f"width: {width}", f"randomSeed: {seed}",
                                                        f"message_length: {message_length}"))
        latent = z.float().cpu().contiguous()
        return ({"samples": latent}, latent[0])


class GSKSamplerAdvanced:
    """nodes.py:167-207 "GS KSamplerAdvanced": KSamplerAdvanced whose initial noise is the GS latent when add_GS_noise is enabled.  Needs ComfyUI."""

    @classmethod
    def INPUT_TYPES(cls):
        import comfy.samplers
        return {"required": {
            "model": ("MODEL",),
            "add_GS_noise": (["enable", "disable"],),
            "add_noise": (["disable", "enable"],),
            "noise_seed": ("INT", {"default": 42, "min": 0, "max": 0xffffffffffffffff}),
            "steps": ("INT", {"default": 20, "min": 1, "max": 10000}),
            "cfg": ("FLOAT", {"default": 8.0, "min": 0.0, "max": 100.0, "step": 0.1, "round": 0.01}),
            "sampler_name": (comfy.samplers.KSampler.SAMPLERS,),
            "scheduler": (comfy.samplers.KSampler.SCHEDULERS,),
            "positive": ("CONDITIONING",),
            "negative": ("CONDITIONING",),
            "latent_image": ("LATENT",),
            "GS_latent_noise": ("LATENT",),
            "start_at_step": ("INT", {"default": 0, "min": 0, "max": 10000}),
            "end_at_step": ("INT", {"default": 10000, "min": 0, "max": 10000}),
            "return_with_leftover_noise": (["disable", "enable"],),
        }}

    RETURN_TYPES = ("LATENT",)
    FUNCTION = "sample"
    CATEGORY = "GSWatermark-lthero/sampling"

    def sample(self, model, add_GS_noise, add_noise, noise_seed, steps, cfg, sampler_name, scheduler, positive, negative, latent_image, GS_latent_noise,
               start_at_step, end_at_step, return_with_leftover_noise, denoise=1.0):
        import comfy.sample
        import comfy.utils
        import latent_preview
        x = latent_image["samples"]
        no_noise = add_noise == "disable"
        if add_GS_noise == "enable":
            noise = GS_latent_noise["samples"]
        elif no_noise:
            noise = torch.zeros(x.size(), dtype=x.dtype, layout=x.layout, device="cpu")
        else:
            noise = comfy.sample.prepare_noise(x, noise_seed, latent_image.get("batch_index"))
        out = dict(latent_image)
        out["samples"] = comfy.sample.sample(model, noise, steps, cfg, sampler_name, scheduler, positive, negative, x, denoise=denoise,
                                             disable_noise=no_noise, start_step=start_at_step, last_step=end_at_step,
                                             force_full_denoise=return_with_leftover_noise != "enable", noise_mask=latent_image.get("noise_mask"),
                                             callback=latent_preview.prepare_callback(model, steps), disable_pbar=not comfy.utils.PROGRESS_BAR_ENABLED,
                                             seed=noise_seed)
        return (out,)


NODE_CLASS_MAPPINGS = {"Lthero_GSLatent": GSLatent, "Lthero_GS_KSamplerAdvanced": GSKSamplerAdvanced}
NODE_DISPLAY_NAME_MAPPINGS = {"Lthero_GSLatent": "GS Latent Noise", "Lthero_GS_KSamplerAdvanced": "GS KSamplerAdvanced"}
