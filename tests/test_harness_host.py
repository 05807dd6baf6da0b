"""CPU: host side of the extract.py twin -- CLI flags/defaults, result.txt text, image loading (extract.py:31-37,165-211)."""
import io
import types

import numpy as np
import pytest
import torch

import gswm_amd
from gswm_amd import extract as E
from gswm_amd import vae as V, unet as U, pipeline as P


def test_cli_flags_and_defaults_match_reference():
    p = E.build_parser()
    a = p.parse_args(["--key_hex", "ab" * 32, "--nonce_hex", "", "--original_message_hex", "6c74"])
    assert (a.model_id, a.num_inference_steps, a.scheduler, a.l, a.width, a.height, a.message_length) == \
        ("stabilityai/stable-diffusion-2-1-base", 30, "DDIM", 1, 1024, 1024, 1024)          # extract.py:183-195
    assert a.images_directory_path == "" and a.single_image_path == "" and a.is_traverse_subdirectories == 0
    with pytest.raises(SystemExit):
        p.parse_args([])                                                                     # the three hex flags are required


def test_write_batch_info_text(tmp_path):
    args = types.SimpleNamespace(key_hex="k", nonce_hex="n", original_message_hex="6c", num_inference_steps=30, scheduler="DDIM")
    f = tmp_path / "result.txt"
    with open(f, "a") as fh:
        E.write_batch_info(fh, args)
    lines = f.read_text().splitlines()
    assert lines[0] == "=" * 40 + "Batch Info" + "=" * 40
    assert lines[1].startswith("Time,") and lines[2:7] == ["key_hex,k", "nonce_hex,n", "original_message_hex,6c", "num_inference_steps,30", "scheduler,DDIM"]
    assert lines[7] == "=" * 40 + "Batch Start" + "=" * 40


def test_load_image_matches_pil_lanczos(tmp_path):
    from PIL import Image
    rng = np.random.RandomState(0)
    img = Image.fromarray(rng.randint(0, 256, (40, 56, 3), dtype=np.uint8))
    p = tmp_path / "a.png"
    img.save(p)
    t = E.load_image(str(p), [32, 24])                                                       # (width, height) like extract.py:63
    assert t.shape == (1, 3, 24, 32) and t.dtype == torch.float32 and 0 <= t.min() and t.max() <= 1
    ref = np.asarray(img.convert("RGB").resize((32, 24), Image.Resampling.LANCZOS), dtype=np.float32) / 255.0
    np.testing.assert_array_equal(t[0].permute(1, 2, 0).numpy(), ref)
    assert E.load_image(str(p)).shape == (1, 3, 40, 56)
    assert E.load_image(str(p), 16).shape == (1, 3, 16, 16)


def test_vae_and_unet_shapes_and_parameter_counts():
    with torch.device("meta"):
        v, u = V.AutoencoderKL(), U.UNet2DCondition()
    assert sum(p.numel() for p in v.parameters()) == 83_653_863           # SD AutoencoderKL
    assert sum(p.numel() for p in u.parameters()) == 865_910_724           # SD 2.1 UNet2DConditionModel
    with torch.device("meta"):
        u15 = U.UNet2DCondition.sd15()
    assert sum(p.numel() for p in u15.parameters()) == 859_520_964         # SD 1.5 UNet2DConditionModel (BASELINE config 5)
    assert [b.attentions[0].transformer_blocks[0].attn1.heads for b in u15.down_blocks[:3]] == [8, 8, 8]
    assert u15.down_blocks[0].attentions[0].transformer_blocks[0].attn2.to_k.in_features == 768
    small = V.synthetic_init_(V.AutoencoderKL(block_out_channels=(32, 32, 64, 64)), 0)
    x = torch.rand(2, 3, 64, 48)
    z = V.img_to_latents(x, small)
    assert z.shape == (2, 4, 8, 6)
    y = V.latents_to_img(z, small)
    assert y.shape == x.shape and 0 <= y.min() and y.max() <= 1


def test_jpeg_roundtrip_is_pil_quality_save():
    from PIL import Image
    x = torch.rand(2, 3, 32, 32)
    y = P.jpeg_roundtrip_pil(x, 10)            # the host-side checker; the device path is tested in test_gpu_harness.py
    assert y.shape == x.shape
    buf = io.BytesIO()
    Image.fromarray(((x[0].clamp(0, 1) * 255).round().to(torch.uint8)).permute(1, 2, 0).numpy()).save(buf, format="JPEG", quality=10)
    buf.seek(0)
    ref = np.asarray(Image.open(buf).convert("RGB"), dtype=np.float32) / 255.0
    np.testing.assert_allclose(y[0].permute(1, 2, 0).numpy(), ref, atol=1e-6)


def test_unknown_scheduler_raises_like_reference():
    args = types.SimpleNamespace(scheduler="Euler", num_inference_steps=5)
    with pytest.raises(ValueError, match="Please choose 'DPMs' or 'DDIM'"):
        E._scheduler_steps(args, types.SimpleNamespace(prediction_type="epsilon"))
