"""CPU: host side of the extract.py twin -- CLI flags/defaults, result.txt text, image loading (extract.py:31-37,165-211)."""
import io
import os
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


# ---------------------------------------------------------------------------------------------------------------------
# model loading: fail closed without a checkpoint, build the modules from the checkpoint's own config files
# ---------------------------------------------------------------------------------------------------------------------
def write_tiny_checkpoint(root, *, dtype=torch.float32, fp16_names=False):
    """A complete diffusers-layout directory with a two-level UNet, a small VAE, a scheduler config, a CLIP tokenizer and a one-layer
    text encoder -- every file `Models` reads, at a size the CPU handles in a second."""
    import json, os
    from safetensors.torch import save_file
    from test_text_host import _write_tokenizer
    from gswm_amd import text as T
    ucfg = {"in_channels": 4, "out_channels": 4, "block_out_channels": [64, 128], "layers_per_block": 2, "cross_attention_dim": 96,
            "attention_head_dim": [1, 2], "down_block_types": ["CrossAttnDownBlock2D", "DownBlock2D"], "norm_num_groups": 32, "act_fn": "silu"}
    vcfg = {"block_out_channels": [32, 64], "latent_channels": 4, "layers_per_block": 2, "norm_num_groups": 32}
    scfg = {"num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear", "steps_offset": 1,
            "set_alpha_to_one": False, "prediction_type": "v_prediction"}
    unet = U.synthetic_init_(E._unet_from_config(ucfg), 3).to(dtype)
    vae = V.synthetic_init_(E._vae_from_config(vcfg), 4).to(dtype)
    name = "diffusion_pytorch_model.fp16.safetensors" if fp16_names else "diffusion_pytorch_model.safetensors"
    for sub, mod, cfg in (("unet", unet, ucfg), ("vae", vae, vcfg)):
        os.makedirs(os.path.join(root, sub))
        save_file({k: v.contiguous() for k, v in mod.state_dict().items()}, os.path.join(root, sub, name))
        with open(os.path.join(root, sub, "config.json"), "w") as f:
            json.dump(cfg, f)
    os.makedirs(os.path.join(root, "scheduler"))
    with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
        json.dump(scfg, f)
    vocab = _write_tokenizer(os.path.join(root, "tokenizer"), "!")
    tcfg = {"vocab_size": len(vocab), "hidden_size": 96, "intermediate_size": 192, "num_hidden_layers": 1, "num_attention_heads": 2,
            "max_position_embeddings": 77, "hidden_act": "gelu", "layer_norm_eps": 1e-5}
    torch.manual_seed(5)
    enc = T.ClipTextEncoder(tcfg).to(dtype)
    os.makedirs(os.path.join(root, "text_encoder"))
    save_file({k: v.contiguous() for k, v in enc.state_dict().items()}, os.path.join(root, "text_encoder", "model.safetensors"))
    with open(os.path.join(root, "text_encoder", "config.json"), "w") as f:
        json.dump(tcfg, f)
    return unet, vae, enc


def test_models_fail_closed_without_checkpoint(monkeypatch):
    monkeypatch.delenv("GSW_ALLOW_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setattr(E, "ALLOW_SYNTHETIC_WEIGHTS", False)
    with pytest.raises(FileNotFoundError, match="allow_synthetic_weights"):
        E.Models("stabilityai/stable-diffusion-2-1-base", device="cpu", dtype=torch.float32)
    a = E.build_parser().parse_args(["--key_hex", "00", "--nonce_hex", "", "--original_message_hex", "6c"])
    assert a.allow_synthetic_weights is False and not E._synthetic_allowed(a)
    a = E.build_parser().parse_args(["--key_hex", "00", "--nonce_hex", "", "--original_message_hex", "6c", "--allow_synthetic_weights"])
    assert E._synthetic_allowed(a)
    monkeypatch.setenv("GSW_ALLOW_SYNTHETIC_WEIGHTS", "1")
    assert E._synthetic_allowed()


@pytest.mark.parametrize("fp16_names", [False, True])
def test_models_load_tiny_checkpoint_from_its_config_files(tmp_path, fp16_names):
    root = str(tmp_path / "ckpt")
    unet, vae, enc = write_tiny_checkpoint(root, fp16_names=fp16_names)
    m = E.Models(root, device="cpu", dtype=torch.float32)
    assert not m.synthetic and m.prediction_type == "v_prediction" and m.ctx_dim == 96
    assert [b.resnets[0].conv1.out_channels for b in m.unet.down_blocks] == [64, 128]
    assert m.unet.down_blocks[0].attentions[0].transformer_blocks[0].attn1.heads == 1 and m.unet.down_blocks[1].attentions is None
    assert m.unet.mid_block.attentions[0].transformer_blocks[0].attn1.heads == 2
    for mine, ref in ((m.unet, unet), (m.vae, vae)):
        sd = ref.state_dict()
        assert all(torch.equal(v, sd[k]) for k, v in mine.state_dict().items())
    assert m.schedule_kwargs() == dict(num_train_timesteps=1000, steps_offset=1, set_alpha_to_one=False, beta_start=0.00085, beta_end=0.012)
    sched = E._scheduler_steps(types.SimpleNamespace(scheduler="DDIM", num_inference_steps=10), m)
    assert sched.prediction_type == "v_prediction" and len(sched.inversion()) == 10
    # the context of the empty prompt comes from the directory's tokenizer + text encoder
    from gswm_amd import text as T
    ids = T.ClipTokenizer.from_dir(root + "/tokenizer")([""])
    assert m.ctx_empty.shape == (1, 77, 96) and torch.allclose(m.ctx_empty, enc(ids), atol=1e-6)
    # one CPU forward of the loaded UNet (plain-torch module path): the config-built module is wired consistently
    y = m.unet(torch.randn(1, 4, 16, 16), torch.tensor([5]), m.ctx_empty)
    assert y.shape == (1, 4, 16, 16) and torch.isfinite(y).all()


def test_unsupported_configs_are_refused():
    with pytest.raises(ValueError, match="layers_per_block"):
        E._unet_from_config({"layers_per_block": 3})
    with pytest.raises(ValueError, match="beta_schedule"):
        m = types.SimpleNamespace(scheduler={"beta_schedule": "linear"})
        E.Models.schedule_kwargs(m)


def test_plan_follows_the_reference_visiting_order(tmp_path):
    """extract.py:120-132: for every os.walk level a banner, then each of its sub-directories; images of a directory are *.png then *.jpg."""
    root = tmp_path / "r"
    for d in ("a", "b", "a/x"):
        (root / d).mkdir(parents=True)
    for f in ("a/1.jpg", "a/0.png", "a/x/2.png", "b/note.txt"):
        (root / f).write_bytes(b"")
    args = types.SimpleNamespace(images_directory_path=str(root), is_traverse_subdirectories=1)
    script = E._plan(args)
    import os
    want = []
    for here, subdirs, _ in os.walk(str(root)):
        want.append(("banner", here))
        want += [("job", os.path.join(here, d)) for d in subdirs]
    assert [(k, x if k == "banner" else x.path) for k, x in script] == want
    ja = next(x for k, x in script if k == "job" and x.path.endswith("/a"))
    assert [os.path.basename(f) for f in ja.files] == ["0.png", "1.jpg"]
    single = E._plan(types.SimpleNamespace(images_directory_path=str(root / "a"), is_traverse_subdirectories=0))
    assert len(single) == 1 and single[0][0] == "job" and single[0][1].path == str(root / "a")


def test_report_text_with_injected_outcomes(tmp_path, capsys):
    """The writer alone (no device): result.txt of a directory and the roll-up line in its parent, byte for byte (extract.py:139-163)."""
    d = tmp_path / "p" / "set"
    d.mkdir(parents=True)
    for f in ("a.png", "b.png", "c.jpg"):
        (d / f).write_bytes(b"")
    args = types.SimpleNamespace(key_hex="k", nonce_hex="n", original_message_hex="6c74", num_inference_steps=30, scheduler="DDIM", model_id=str(tmp_path))
    job = E._DirJob(str(d))
    bits_ok = "0110110001110100"
    bits_bad = "0110110001110111"
    job.outcome = {str(d / "a.png"): bits_ok, str(d / "b.png"): ValueError("boom"), str(d / "c.jpg"): bits_bad}
    E._report(job, args, synthetic=False)
    out = capsys.readouterr().out
    _, acc_bad = E.calculate_bit_accuracy("6c74", bits_bad)
    assert out == (f"a.png\nOriginal Message: {bits_ok} \nExtracted Message: {bits_ok}\nBit Accuracy: 1.0\n\n"
                   f"Error processing {d / 'b.png'}: boom\n\n"
                   f"c.jpg\nOriginal Message: {bits_ok} \nExtracted Message: {bits_bad}\nBit Accuracy: {acc_bad}\n\n")
    lines = (d / "result.txt").read_text().split("\n")
    bar = "=" * 40
    assert lines[8:] == ["a.png, Bit Accuracy, 1.0", f"Error processing {d / 'b.png'}: boom", f"c.jpg, Bit Accuracy, {acc_bad}",
                         f"Average Bit Accuracy, {(1.0 + acc_bad) / 2}", "", f"{bar}Batch End{bar}", ""]
    assert (tmp_path / "p" / "result.txt").read_text() == f"set, Average Bit Accuracy, {(1.0 + acc_bad) / 2}\n"
    # synthetic weights: marker line after the header, no roll-up
    (tmp_path / "p" / "result.txt").unlink()
    (d / "result.txt").unlink()
    E._report(job, args, synthetic=True)
    lines = (d / "result.txt").read_text().split("\n")
    assert lines[8].startswith("SYNTHETIC WEIGHTS,") and lines[9] == "a.png, Bit Accuracy, 1.0"
    assert not (tmp_path / "p" / "result.txt").exists()


def test_recover_many_isolates_failures_per_image(tmp_path, monkeypatch):
    """A file that does not decode fails alone; a device batch that raises is redone image by image so each reports its own error."""
    from PIL import Image
    d = tmp_path / "s"
    d.mkdir()
    for i in range(5):
        Image.fromarray(np.full((8, 8, 3), i, dtype=np.uint8)).save(d / f"{i}.png")
    (d / "zz.png").write_bytes(b"junk")
    job = E._DirJob(str(d))
    calls = []

    def fake_invert(arrs, args, device="cuda"):
        calls.append(len(arrs))
        if any(int(a[0, 0, 0]) == 3 for a in arrs):
            raise RuntimeError("poisoned image")
        return torch.tensor([int(a[0, 0, 0]) for a in arrs])

    monkeypatch.setattr(E, "invert_decoded_images", fake_invert)
    monkeypatch.setattr(E, "recover_exactracted_message_batch", lambda lat, args: [f"bits{int(v)}" for v in lat])
    E._recover_many([(job, f) for f in job.files], None, batch_size=4)
    got = {os.path.basename(k): v for k, v in job.outcome.items()}
    assert isinstance(got["zz.png"], Exception) and isinstance(got["3.png"], RuntimeError) and str(got["3.png"]) == "poisoned image"
    assert [got[f"{i}.png"] for i in (0, 1, 2, 4)] == ["bits0", "bits1", "bits2", "bits4"]
    good = [os.path.basename(f) for f in job.files if not f.endswith("zz.png")]          # glob order is the file system's
    want = [4, 1, 1, 1, 1, 1] if "3.png" in good[:4] else [4, 1, 1]    # a batch that raises is redone singly; the other batch runs once
    assert calls == want


def test_comfy_node_interface_matches_the_reference_workflow_names():
    """nodes.py:209-252: node keys, display names, input names / types / defaults, outputs, categories -- what a saved workflow JSON refers to."""
    from gswm_amd import comfy as C
    assert set(C.NODE_CLASS_MAPPINGS) == {"Lthero_GSLatent", "Lthero_GS_KSamplerAdvanced"}
    assert C.NODE_DISPLAY_NAME_MAPPINGS == {"Lthero_GSLatent": "GS Latent Noise", "Lthero_GS_KSamplerAdvanced": "GS KSamplerAdvanced"}
    n = C.GSLatent
    req = n.INPUT_TYPES()["required"]
    assert list(req) == ["use_seed", "seed", "width", "height", "key", "nonce", "message", "message_length", "batch_size"]
    assert req["message_length"] == ("INT", {"default": -1, "min": 32, "max": 1024, "step": 32}) and req["batch_size"][1]["max"] == 64
    assert req["key"][1]["default"] == "5822ff9cce6772f714192f43863f6bad1bf54b78326973897e6b66c3186b77a7" and req["message"][1]["default"] == "lthero"
    assert n.RETURN_TYPES == ("LATENT", "IMAGE") and n.FUNCTION == "create_gs_latents" and n.CATEGORY == "GSWatermark-lthero/latent/noise"
    k = C.GSKSamplerAdvanced
    assert k.RETURN_TYPES == ("LATENT",) and k.FUNCTION == "sample" and k.CATEGORY == "GSWatermark-lthero/sampling"


# ---------------------------------------------------------------------------------------------------------------------
# real-checkpoint readiness: hub ids through the local Hugging Face cache, weight file formats, SD 1.5's layout, config validation
# ---------------------------------------------------------------------------------------------------------------------
def _hub_cache_with(tmp_path, monkeypatch, repo_id, build):
    """<cache>/models--org--name/{refs/main, snapshots/<rev>/...} as huggingface_hub lays it out; `build(snapshot_dir)` fills the snapshot"""
    import os
    cache = tmp_path / "hf_home" / "hub"
    rev = "0123456789abcdef0123456789abcdef01234567"
    snap = cache / ("models--" + repo_id.replace("/", "--")) / "snapshots" / rev
    os.makedirs(snap.parent.parent / "refs")
    (snap.parent.parent / "refs" / "main").write_text(rev)
    out = build(str(snap))
    for var in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "DIFFUSERS_CACHE", "XDG_CACHE_HOME"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf_home"))
    return str(snap), out


def test_hub_id_resolves_through_the_local_hf_cache(tmp_path, monkeypatch):
    """the reference's default --model_id (extract.py:183) on a machine that has run the reference: found under $HF_HOME/hub, nothing downloaded"""
    from gswm_amd import checkpoint as C
    monkeypatch.setattr(E, "_MODEL_CACHE", {})
    snap, (unet, vae, enc) = _hub_cache_with(tmp_path, monkeypatch, "stabilityai/stable-diffusion-2-1-base", lambda d: write_tiny_checkpoint(d))
    assert C.resolve_model_dir("stabilityai/stable-diffusion-2-1-base") == snap
    assert C.resolve_model_dir("stabilityai/another-model") is None and C.resolve_model_dir("not a repo id") is None
    assert not E._no_checkpoint("stabilityai/stable-diffusion-2-1-base") and E._no_checkpoint("runwayml/stable-diffusion-v1-5")
    m = E.Models("stabilityai/stable-diffusion-2-1-base", device="cpu", dtype=torch.float32)
    assert not m.synthetic and m.model_id == snap and m.model_name == "stabilityai/stable-diffusion-2-1-base"
    sd = unet.state_dict()
    assert all(torch.equal(v, sd[k]) for k, v in m.unet.state_dict().items())
    # HF_HUB_CACHE wins over HF_HOME, and a missing id says where it looked
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "elsewhere"))
    assert C.resolve_model_dir("stabilityai/stable-diffusion-2-1-base") == snap           # still found through HF_HOME (every root is searched)
    assert C.hub_cache_dirs()[0] == str(tmp_path / "elsewhere")
    with pytest.raises(FileNotFoundError, match="models--runwayml--stable-diffusion-v1-5"):
        E.Models("runwayml/stable-diffusion-v1-5", device="cpu", dtype=torch.float32)


@pytest.mark.parametrize("fmt", ["bin", "bin_fp16", "sharded_safetensors", "sharded_bin"])
def test_weight_file_formats(tmp_path, fmt):
    """.bin (torch.save) and sharded checkpoints (index.json + shards) load to the same modules as the single safetensors file"""
    import json, os
    from safetensors.torch import load_file, save_file
    root = str(tmp_path / "ckpt")
    unet, vae, enc = write_tiny_checkpoint(root)
    for sub, stem in (("unet", "diffusion_pytorch_model"), ("vae", "diffusion_pytorch_model"), ("text_encoder", "model")):
        src = os.path.join(root, sub, stem + ".safetensors")
        sd = load_file(src)
        os.remove(src)
        if fmt == "bin":
            torch.save(sd, os.path.join(root, sub, ("pytorch_model" if sub == "text_encoder" else stem) + ".bin"))
        elif fmt == "bin_fp16":
            torch.save(sd, os.path.join(root, sub, stem + ".fp16.bin"))
        else:
            ext = ".safetensors" if fmt == "sharded_safetensors" else ".bin"
            keys = sorted(sd)
            parts = [keys[: len(keys) // 3], keys[len(keys) // 3: 2 * len(keys) // 3], keys[2 * len(keys) // 3:]]
            wm = {}
            for i, ks in enumerate(parts):
                name = f"{stem}-{i + 1:05d}-of-00003{ext}"
                part = {k: sd[k].contiguous() for k in ks}
                save_file(part, os.path.join(root, sub, name)) if ext == ".safetensors" else torch.save(part, os.path.join(root, sub, name))
                wm.update({k: name for k in ks})
            with open(os.path.join(root, sub, stem + ext + ".index.json"), "w") as f:
                json.dump({"metadata": {}, "weight_map": wm}, f)
    m = E.Models(root, device="cpu", dtype=torch.float32)
    for mine, ref in ((m.unet, unet), (m.vae, vae)):
        sd = ref.state_dict()
        assert all(torch.equal(v, sd[k]) for k, v in mine.state_dict().items())
    from gswm_amd import text as T
    assert torch.allclose(m.ctx_empty, enc(T.ClipTokenizer.from_dir(root + "/tokenizer")([""])), atol=1e-6)
    # a shard named by the index but missing on disk is an error, not a partial load
    if fmt.startswith("sharded"):
        victim = [f for f in os.listdir(os.path.join(root, "unet")) if "-00002-" in f][0]
        os.remove(os.path.join(root, "unet", victim))
        with pytest.raises(FileNotFoundError, match="00002"):
            E.Models(root, device="cpu", dtype=torch.float32)


def _sd_shaped_config(kind):
    """unet/config.json of stabilityai/stable-diffusion-2-1-base resp. runwayml/stable-diffusion-v1-5 as published (widths divided by 5 for the CPU)"""
    base = {"_class_name": "UNet2DConditionModel", "_diffusers_version": "0.10.0", "act_fn": "silu", "center_input_sample": False, "downsample_padding": 1,
            "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 4, "out_channels": 4, "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05,
            "norm_num_groups": 32, "block_out_channels": [64, 128, 256, 256], "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
            "up_block_types": ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3}
    if kind == "sd21":
        base.update({"attention_head_dim": [1, 2, 4, 4], "cross_attention_dim": 1024, "dual_cross_attention": False, "only_cross_attention": False,
                     "sample_size": 64, "use_linear_projection": True, "upcast_attention": True, "num_class_embeds": None, "class_embed_type": None,
                     "mid_block_type": "UNetMidBlock2DCrossAttn", "resnet_time_scale_shift": "default"})
    else:
        base.update({"attention_head_dim": 8, "cross_attention_dim": 768, "sample_size": 64})
    return base


@pytest.mark.parametrize("kind", ["sd21", "sd15"])
def test_sd_shaped_unet_configs_load_in_diffusers_key_layout(tmp_path, kind):
    """the published config.json shapes of SD 2.1-base (linear proj_in / proj_out, per-block head counts, upcast_attention) and SD 1.5 (8 heads
    everywhere, proj_in / proj_out stored as 1x1 convolutions [C, C, 1, 1]) round-trip through the loader and compute the same forward"""
    import json, os
    from safetensors.torch import save_file
    cfg = _sd_shaped_config(kind)
    ref = U.synthetic_init_(E._unet_from_config(cfg), 11)
    heads = [b.attentions[0].transformer_blocks[0].attn1.heads for b in ref.down_blocks if b.attentions is not None]
    assert heads == ([1, 2, 4] if kind == "sd21" else [8, 8, 8])
    sd = {k: v.contiguous() for k, v in ref.state_dict().items()}
    if kind == "sd15":
        for k in list(sd):
            if k.endswith(("proj_in.weight", "proj_out.weight")):
                sd[k] = sd[k][:, :, None, None].contiguous()          # diffusers < use_linear_projection: Conv2d(C, C, 1)
    root = str(tmp_path / kind)
    os.makedirs(os.path.join(root, "unet"))
    save_file(sd, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    with open(os.path.join(root, "unet", "config.json"), "w") as f:
        json.dump(cfg, f)
    mine = E._unet_from_config(E._read_json(os.path.join(root, "unet", "config.json")))
    U.load_diffusers_state_dict(mine, root)
    rs = ref.state_dict()
    assert all(torch.equal(v, rs[k]) for k, v in mine.state_dict().items())
    x, t, c = torch.randn(1, 4, 16, 16), torch.tensor([7]), torch.randn(1, 77, cfg["cross_attention_dim"])
    with torch.no_grad():
        assert torch.equal(mine(x, t, c), ref(x, t, c))
    # a tensor too many, or one missing, is refused (strict both ways)
    sd["extra.weight"] = torch.zeros(1)
    save_file(sd, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    with pytest.raises(RuntimeError, match="unexpected"):
        U.load_diffusers_state_dict(E._unet_from_config(cfg), root)


def test_config_entries_outside_the_implemented_space_are_refused_by_name():
    from gswm_amd import checkpoint as C
    ok = _sd_shaped_config("sd21")
    C.validate_unet_config(ok)
    for k, v in (("class_embed_type", "timestep"), ("addition_embed_type", "text_time"), ("dual_cross_attention", True), ("transformer_layers_per_block", 2),
                 ("resnet_time_scale_shift", "scale_shift"), ("mid_block_type", "UNetMidBlock2DSimpleCrossAttn"), ("only_cross_attention", True),
                 ("time_embedding_type", "fourier"), ("norm_eps", 1e-6), ("encoder_hid_dim", 1024), ("conv_out_kernel", 1)):
        with pytest.raises(ValueError, match=k):
            C.validate_unet_config(dict(ok, **{k: v}))
    with pytest.raises(ValueError, match="down block type"):
        C.validate_unet_config(dict(ok, down_block_types=["SimpleCrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]))
    with pytest.raises(ValueError, match="mirror"):
        C.validate_unet_config(dict(ok, up_block_types=["CrossAttnUpBlock2D"] * 4))
    C.validate_scheduler_config({"prediction_type": "v_prediction", "timestep_spacing": "leading", "clip_sample": False})
    for k, v in (("prediction_type", "sample"), ("timestep_spacing", "trailing"), ("clip_sample", True), ("rescale_betas_zero_snr", True), ("beta_schedule", "linear")):
        with pytest.raises(ValueError, match=k if k != "clip_sample" else "clip_sample"):
            C.validate_scheduler_config({k: v})


def test_full_precision_weights_win_over_an_fp16_variant_in_the_same_directory(tmp_path):
    """a repository that ships `<stem>.bin` (fp32) next to `<stem>.fp16.safetensors`: a load must not silently get the fp16-rounded copy -- full-precision files of
    BOTH formats come before any `.fp16` variant; with only variants present, the safetensors one wins"""
    import os
    from safetensors.torch import save_file
    from gswm_amd import checkpoint as C
    d = tmp_path / "unet"
    os.makedirs(d)
    w = torch.tensor([1.0 + 2.0 ** -20, 3.0])                      # not representable in fp16
    torch.save({"w": w}, str(d / "diffusion_pytorch_model.bin"))
    save_file({"w": w.half()}, str(d / "diffusion_pytorch_model.fp16.safetensors"))
    sd = C.load_component_state_dict(str(tmp_path), "unet")
    assert sd["w"].dtype == torch.float32 and torch.equal(sd["w"], w)
    os.remove(d / "diffusion_pytorch_model.bin")
    torch.save({"w": (w * 2).half()}, str(d / "diffusion_pytorch_model.fp16.bin"))
    sd = C.load_component_state_dict(str(tmp_path), "unet")
    assert sd["w"].dtype == torch.float16 and torch.equal(sd["w"], w.half())
