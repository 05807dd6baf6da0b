"""End-to-end Gaussian-Shading path on one GPU: embed -> (G1) DDIM sampling -> (X2) DDIM inversion -> (X3-X5) extract,
everything resident on the device.

Replaces, for the hot path, the reference's per-image flow (extract.py:46-117: `from_pretrained` per image, `.cpu()` of the
inverted latent, scalar Python loops) and the generation loop of modified_stable_diffusion_gs.pyc: the pipeline object is
built once, a batch of independent images moves through the loops together, and nothing returns to the host but
`B x message_length/8` bytes.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import codec
from .ddim import DDIMSchedule, ddim_invert_extract, ddim_sample, ddim_invert


class GaussianShadingPipeline:
    def __init__(self, eps_model, key: bytes, nonce: bytes, message: bytes, *, height: int = 512, width: int = 512,
                 num_inference_steps: int = 50, dtype: torch.dtype = torch.float16, device="cuda",
                 ctx_uncond: Optional[torch.Tensor] = None, prediction_type: str = "epsilon"):
        if hasattr(eps_model, "prepare_context"):          # a unet.UNet2DCondition: small batches replay a captured HIP graph of the forward (graph.py)
            from .graph import graphed
            eps_model = graphed(eps_model, clone_output=False)      # the loops of ddim.py consume eps in the scheduler-step kernel right away
        self.eps_model = eps_model
        self.key, self.nonce, self.message = key, nonce, message
        self.shape = (4, height // 8, width // 8)
        self.dtype, self.device = dtype, torch.device(device)
        self.schedule = DDIMSchedule(num_inference_steps=num_inference_steps, prediction_type=prediction_type)
        self.ctx_uncond = ctx_uncond
        self.message_length = 8 * len(message)

    # E1-E6
    def embed(self, batch: int, *, seed: int = 0, image_index0: int = 0, fast: bool = True, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        return codec.embed_batch(self.key, self.nonce, self.message, batch, self.shape, u=u, seed=seed, image_index0=image_index0,
                                 dtype=self.dtype, fast=fast, device=self.device)

    # G1
    def _uncond(self, batch: int) -> torch.Tensor:
        """the empty prompt's context for `batch` images: ONE view object per batch size (the eps model's per-context caches live on the tensor object)"""
        if self.ctx_uncond.shape[0] != 1:
            return self.ctx_uncond
        views = self.__dict__.setdefault("_uncond_views", {})
        ent = views.get(batch)
        if ent is None or ent[0] is not self.ctx_uncond:
            ent = views[batch] = (self.ctx_uncond, self.ctx_uncond.expand(batch, -1, -1))
        return ent[1]

    def generate(self, z_T: torch.Tensor, ctx_text: torch.Tensor, guidance_scale: float = 7.5) -> torch.Tensor:
        cu = self._uncond(z_T.shape[0]) if guidance_scale != 1.0 else None
        return ddim_sample(self.eps_model, z_T, ctx_text, self.schedule, ctx_uncond=cu, guidance_scale=guidance_scale)

    # X2 + X3-X5 (prompt "" -> the unconditional context, guidance 1: extract.py:66-69)
    def invert_and_extract(self, x0: torch.Tensor, *, return_latents: bool = False):
        ctx = self._uncond(x0.shape[0])
        return ddim_invert_extract(self.eps_model, x0, ctx, self.schedule, self.key, self.nonce, self.message_length,
                                   return_latents=return_latents)

    def invert(self, x0: torch.Tensor) -> torch.Tensor:
        ctx = self._uncond(x0.shape[0])
        return ddim_invert(self.eps_model, x0, ctx, self.schedule)

    def txt2img(self, ctx_text: torch.Tensor, vae, *, latents: Optional[torch.Tensor] = None, seed: int = 0, image_index0: int = 0,
                guidance_scale: float = 7.5):
        """The call of the reference's generation pipeline (`ModifiedStableDiffusionPipeline.__call__` of modified_stable_diffusion_gs.pyc:
        `prepare_latents(latents=Z_s_T)`, keep a copy as `init_latents`, CFG sampling loop, `decode_latents`) with its return order:
        (images [B,3,H,W] in [0,1], has_nsfw_concept = None (no safety checker here), init_latents).  latents=None embeds a fresh Z_s_T batch
        the size of ctx_text."""
        z_T = self.embed(ctx_text.shape[0], seed=seed, image_index0=image_index0) if latents is None else latents.to(self.device, self.dtype)
        init_latents = z_T.clone()
        x0 = self.generate(z_T, ctx_text, guidance_scale)
        return decode_images(x0, vae), None, init_latents

    def roundtrip(self, batch: int, ctx_text: torch.Tensor, *, seed: int = 0, image_index0: int = 0, guidance_scale: float = 7.5):
        z_T = self.embed(batch, seed=seed, image_index0=image_index0)
        x0 = self.generate(z_T, ctx_text, guidance_scale)
        bits, flags = self.invert_and_extract(x0)
        return z_T, x0, bits, flags


# ---------------------------------------------------------------------------------------------------------------------
# image-level stages around the latent loops (X1 / `decode_image`), and the JPEG distortion of BASELINE config 4
# ---------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def decode_images(latents: torch.Tensor, vae) -> torch.Tensor:
    """`decode_image` + `torch_to_numpy` prefix of modified_stable_diffusion_gs.pyc: [B,4,h,w] -> [B,3,8h,8w] in [0,1]."""
    from .vae import latents_to_img
    return latents_to_img(latents, vae)


@torch.no_grad()
def encode_images(images: torch.Tensor, vae) -> torch.Tensor:
    """extract.py:39-43 on a batch already on the device."""
    from .vae import img_to_latents
    return img_to_latents(images, vae)


def jpeg_roundtrip(images: torch.Tensor, quality: int = 10) -> torch.Tensor:
    """JPEG distortion as the reference's `distortions` tool applies it (distortions:175-184: PIL save(quality=QF) / reload) on a
    [B,3,H,W] tensor in [0,1]: quantise to uint8 like numpy_to_pil, run the libjpeg-exact lossy stages on the device
    (imaging.jpeg_roundtrip), return ToTensor values in the input dtype."""
    from . import imaging
    u8 = imaging.tensor_to_image(images)
    return imaging.jpeg_roundtrip(u8, quality, out="f32").to(images.dtype)


def jpeg_roundtrip_pil(images: torch.Tensor, quality: int = 10) -> torch.Tensor:
    """The same through host-side PIL, one image at a time (what the reference does); kept as the checker for the device path."""
    import io
    import numpy as np
    from PIL import Image
    out = []
    for img in (images.detach().float().clamp(0, 1).cpu() * 255).round().to(torch.uint8):
        buf = io.BytesIO()
        Image.fromarray(img.permute(1, 2, 0).numpy()).save(buf, format="JPEG", quality=int(quality))
        buf.seek(0)
        out.append(torch.from_numpy(np.asarray(Image.open(buf).convert("RGB")).copy()).permute(2, 0, 1))
    return (torch.stack(out).float() / 255.0).to(images.device, images.dtype)
