"""GPU: BASELINE.json's configurations at FULL size on one MI355X -- configs[1] and configs[2] as named, and one rank's shard of configs[3]
(256 images / 8 GPUs = 32) and configs[4] (128 / 8 = 16).  Synthetic weights (no checkpoint is reachable here), so what is gated is what
does not depend on trained weights: the embedded latents against the oracle, the lossless embed -> sample -> invert -> vote round trip
(100 % of the bits), the vote against the oracle on the very same inverted latents, the image stages against their host checkers, and
that NOTHING leaves the hand-written kernels (unet.FALLBACKS stays empty)."""
import numpy as np
import pytest
import torch

import gs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gswm_amd
    from gswm_amd import codec, ddim, pf, pipeline, unet, vae, imaging
    gswm_amd._native.lib()
    return gswm_amd


def _model(G, kind):
    U = G.unet
    m = U.UNet2DCondition() if kind == "sd21" else U.UNet2DCondition.sd15()
    return U.synthetic_init_(m, 0).cuda().half().eval()


@pytest.fixture(scope="module")
def sd21(G):
    return _model(G, "sd21")


@pytest.fixture(scope="module")
def vae(G):
    return G.vae.synthetic_init_(G.vae.AutoencoderKL(), 1).cuda().half().eval()


def _contexts(B, dim):
    g = torch.Generator().manual_seed(1)
    return (torch.randn(1, 77, dim, generator=g).cuda().half(), torch.randn(B, 77, dim, generator=g).cuda().half())


def _oracle_embed(key, nonce, k, seed, index, shape):
    n = int(np.prod(shape))
    u = O.philox_uniform(seed, index, 1, n)[0]
    return O.embed_latent(k, key, nonce, u, shape)


def test_config2_txt2img_batch8_50_steps_embed_only(G, sd21, vae, keys):
    """configs[1]: SD2.1 512x512 txt2img, batch 8, 50 DDIM steps, guidance 7.5, embed-only path: Z_s_T -> latents -> images."""
    key, nonce = keys
    k = O.pad_message("lthero", 32)
    cu, ct = _contexts(8, 1024)
    pipe = G.pipeline.GaussianShadingPipeline(sd21, key, nonce, k, height=512, width=512, num_inference_steps=50, ctx_uncond=cu)
    G.unet.FALLBACKS.clear()
    z32 = G.codec.embed_batch(key, nonce, k, 8, (4, 64, 64), seed=11, dtype=torch.float32, fast=True)
    for b in (0, 7):
        ref = _oracle_embed(key, nonce, k, 11, b, (4, 64, 64))
        assert np.abs(z32[b].cpu().numpy().astype(np.float64) - ref).max() <= 1e-5            # north_star: within 1e-5 of the CPU path
    zT = pipe.embed(8, seed=11)
    assert torch.equal(zT, z32.half())                                                          # the fp16 state is the rounded fp32 embed
    x0 = pipe.generate(zT, ct, 7.5)
    assert x0.shape == (8, 4, 64, 64) and torch.isfinite(x0).all()
    img = G.pipeline.decode_images(x0, vae)
    img2, nsfw, init = pipe.txt2img(ct, vae, latents=zT)                                       # the reference pipeline's call: (images, nsfw, init_latents)
    assert nsfw is None and torch.equal(init, zT) and img2.shape == img.shape
    assert (img2.float() - img.float()).abs().max().item() <= 2e-2                             # same path twice
    assert img.shape == (8, 3, 512, 512) and torch.isfinite(img).all() and 0 <= img.min() and img.max() <= 1
    u8 = G.imaging.tensor_to_image(img)
    assert u8.dtype == torch.uint8 and u8.shape == (8, 512, 512, 3)
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS


def test_config3_embed_invert_extract_batch32_lossless(G, sd21, keys):
    """configs[2]: SD2.1 512x512 embed + 50-step DDIM-inversion extract, batch 32, lossless -- the 100 % bit-accuracy gate -- plus the vote
    checked against the oracle on the very latents the inversion produced."""
    key, nonce = keys
    k = O.pad_message("lthero", 32)
    B = 32
    cu, ct = _contexts(B, 1024)
    pipe = G.pipeline.GaussianShadingPipeline(sd21, key, nonce, k, height=512, width=512, num_inference_steps=50, ctx_uncond=cu)
    G.unet.FALLBACKS.clear()
    zT = pipe.embed(B, seed=2024)
    x0 = pipe.generate(zT, ct, 7.5)
    bits, flags, zi = pipe.invert_and_extract(x0, return_latents=True)
    assert bits.shape == (B, 32) and zi.shape == zT.shape
    m = G.codec.bit_matches(bits, 256, k)
    assert int(m.sum()) == B * 256, f"lossless gate: {int(m.sum())} of {B * 256} bits"
    zi_np = zi.float().cpu().numpy()
    for b in (0, 13, 31):
        assert G.codec.bits_to_str(bits[b].cpu().numpy()) == O.recover_bits(zi_np[b], key, nonce, 256)
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS


def test_config4_shard_jpeg_qf10_batch32(G, sd21, vae, keys):
    """One rank's 32 images of configs[3]: embed -> sample -> VAE decode -> uint8 -> JPEG QF 10 -> VAE encode -> invert -> vote.  The device
    JPEG equals PIL's save / reload on full 512x512 images; the re-encoded latents go through the inversion and the vote (synthetic VAE
    weights are not an autoencoder, so their accuracy is not gated; the latent-level round trip of the same batch is)."""
    key, nonce = keys
    k = O.pad_message("lthero", 32)
    B = 32
    cu, ct = _contexts(B, 1024)
    pipe = G.pipeline.GaussianShadingPipeline(sd21, key, nonce, k, height=512, width=512, num_inference_steps=50, ctx_uncond=cu)
    G.unet.FALLBACKS.clear()
    zT = pipe.embed(B, seed=7, image_index0=32 * 3)                     # rank 3's slice of the global image index space
    x0 = pipe.generate(zT, ct, 7.5)
    lat = []
    for i, chunk in enumerate(x0.split(8)):
        img = G.vae.latents_to_img(chunk, vae)
        u8 = G.imaging.tensor_to_image(img)
        xn = G.imaging.jpeg_roundtrip(u8, 10, out="f16")
        if i == 0:
            chk = G.pipeline.jpeg_roundtrip_pil(img[:2].float(), 10)
            assert torch.equal(((xn[:2].float() + 1.0) * 0.5 * 255.0).round(), (chk * 255.0).round().to(xn.device))
        lat.append(G.vae.normalised_img_to_latents(xn, vae))
    lat = torch.cat(lat)
    assert lat.shape == x0.shape and torch.isfinite(lat).all()
    bits_j, flags_j, zj = pipe.invert_and_extract(lat, return_latents=True)
    assert bits_j.shape == (B, 32) and torch.isfinite(zj).all()
    zj_np = zj.float().cpu().numpy()
    for b in (0, 31):
        assert G.codec.bits_to_str(bits_j[b].cpu().numpy()) == O.recover_bits(zj_np[b], key, nonce, 256)
    bits, flags = pipe.invert_and_extract(x0)
    assert int(G.codec.bit_matches(bits, 256, k).sum()) == B * 256
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS


@pytest.mark.parametrize("msg_bits", [256, 1024])
def test_config5_shard_sd15_768_batch16(G, keys, msg_bits):
    """One rank's 16 images of configs[4]: SD 1.5-shaped UNet (8 heads: head_dim 40 / 80 / 160, context 768) on the 4x96x96 lattice (odd
    tiles: 96-wide rows, 9216 / 2304 / 576 / 144-token sequences), 256-bit message and the 1024-bit one extract.py defaults to."""
    key, nonce = keys
    k = O.pad_message("lthero", msg_bits // 8)
    B = 16
    model = _model(G, "sd15")
    cu, ct = _contexts(B, 768)
    pipe = G.pipeline.GaussianShadingPipeline(model, key, nonce, k, height=768, width=768, num_inference_steps=50, ctx_uncond=cu)
    G.unet.FALLBACKS.clear()
    z32 = G.codec.embed_batch(key, nonce, k, 2, (4, 96, 96), seed=5, dtype=torch.float32, fast=True)
    assert np.abs(z32[1].cpu().numpy().astype(np.float64) - _oracle_embed(key, nonce, k, 5, 1, (4, 96, 96))).max() <= 1e-5
    zT, x0, bits, flags = pipe.roundtrip(B, ct, seed=5, guidance_scale=7.5)
    assert x0.shape == (B, 4, 96, 96) and bits.shape == (B, msg_bits // 8)
    assert int(G.codec.bit_matches(bits, msg_bits, k).sum()) == B * msg_bits
    assert G.unet.FALLBACKS == {}, G.unet.FALLBACKS
    del model
    torch.cuda.empty_cache()
