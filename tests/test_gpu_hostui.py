"""GPU: the host-UI adapters -- ComfyUI node classes and the WebUI ImageRNG replacement -- against the oracle's codec and the
single-image twins, which are themselves pinned to vectors produced by the reference's nodes.py / gs_insert.py."""
import types

import numpy as np
import pytest
import torch

import gs_oracle as O
from conftest import README_KEY, README_NONCE

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gswm_amd
    from gswm_amd import comfy, webui
    return gswm_amd


def test_comfy_gslatent_batch_matches_the_single_image_twin(G, tmp_path):
    C = G.comfy
    node = C.GSLatent()
    out, first = node.create_gs_latents(README_KEY, README_NONCE, "lthero", 3, 1, 42, 768, 512, -1, log_path=str(tmp_path / "info.txt"))
    lat = out["samples"]
    assert lat.shape == (3, 4, 64, 96) and lat.dtype == torch.float32 and not lat.is_cuda and torch.equal(first, lat[0])
    one = C.gs_watermark_init_noise(README_KEY, README_NONCE, "cpu", "lthero", 1, 42, 768, 512, -1, log_path=None)
    assert torch.equal(lat[0], one) and torch.equal(lat[1], one) and torch.equal(lat[2], one)          # seeded: the whole batch is one lattice
    assert (tmp_path / "info.txt").read_text().count("----------------------") == 1
    # unseeded: the global numpy stream advances image by image, exactly as three calls of the single-image function would draw from it
    np.random.seed(7)
    out, _ = node.create_gs_latents(README_KEY, README_NONCE, "abc", 2, 0, 0, 512, 512, 256, log_path=None)
    after_batch = np.random.get_state()[2], np.random.get_state()[1][:4].copy()
    np.random.seed(7)
    a = C.gs_watermark_init_noise(README_KEY, README_NONCE, "cpu", "abc", 0, 0, 512, 512, 256, log_path=None)
    b = C.gs_watermark_init_noise(README_KEY, README_NONCE, "cpu", "abc", 0, 0, 512, 512, 256, log_path=None)
    assert torch.equal(out["samples"][0], a) and torch.equal(out["samples"][1], b)
    assert after_batch[0] == np.random.get_state()[2] and np.array_equal(after_batch[1], np.random.get_state()[1][:4])


def test_webui_image_rng_first_noise_is_the_watermarked_batch(G, tmp_path):
    W = G.webui
    fake = types.SimpleNamespace(ImageRNG=object)
    prev = W.install(fake, message="lthero", key_hex=README_KEY, nonce_hex=README_NONCE, use_randomSeed=1, randomSeed=42, use_repeat=1)
    assert prev is object and fake.ImageRNG is W.GaussianShadingImageRNG
    rng = fake.ImageRNG((4, 64, 64), seeds=[11, 12])
    import os
    cwd = os.getcwd()
    os.chdir(tmp_path)                              # the script appends ./info_data.txt
    try:
        x = rng.next()
    finally:
        os.chdir(cwd)
    assert x.shape == (2, 4, 64, 64) and x.is_cuda and x.dtype == torch.float32
    key, nonce = bytes.fromhex(README_KEY), bytes.fromhex(README_NONCE)
    k = O.pad_message("lthero", 8) * 4                                            # use_repeat: 8 bytes, four times
    u = np.random.RandomState(seed=42).uniform(0, 1, 2 * 16384).reshape(2, -1)
    for b in range(2):
        ref = O.embed_latent(k, key, nonce, u[b], (4, 64, 64))
        assert np.abs(x[b].cpu().numpy().astype(np.float64) - ref).max() <= 1e-6
    txt = (tmp_path / "info_data.txt").read_text().splitlines()
    assert txt[1] == f"key: {README_KEY}" and txt[3] == "randomSeed: 42" and txt[4] == f"message: {k.hex()}"
    y = rng.next()                                                               # afterwards: ordinary per-seed noise
    assert y.shape == (2, 4, 64, 64) and torch.equal(y[0].cpu(), torch.randn((4, 64, 64), generator=torch.Generator("cpu").manual_seed(11)))
    with pytest.raises(TypeError):
        W.install(fake, nonsense=1)
