"""Where the weights of `--model_id` live and how they are read (row H1 of SURVEY.md section 8a; the reference: `StableDiffusionPipeline.from_pretrained(
args.model_id, ...)` at extract.py:56-60 with the default `stabilityai/stable-diffusion-2-1-base` of extract.py:183, which diffusers resolves
through the local Hugging Face cache when the snapshot has been downloaded before).

There is no network here and no diffusers: a hub id is looked up in the SAME cache layout huggingface_hub writes
(`<cache>/models--<org>--<name>/snapshots/<revision>/`, revision from `refs/main` when present), so a machine that has already run the
reference finds its checkpoint without a path change.  Weight files are read in diffusers' own preference order: single safetensors file, the fp16
variant, sharded safetensors (`*.safetensors.index.json`), then the pickle formats (`.bin`, torch.load with weights_only=True).

Pure host code: no GPU, no oracle."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import torch


def hub_cache_dirs() -> List[str]:
    """Cache roots in huggingface_hub's precedence: HF_HUB_CACHE, HUGGINGFACE_HUB_CACHE, $HF_HOME/hub, ~/.cache/huggingface/hub (+ diffusers' old
    DIFFUSERS_CACHE)."""
    out = []
    for var in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE", "DIFFUSERS_CACHE"):
        if os.environ.get(var):
            out.append(os.environ[var])
    if os.environ.get("HF_HOME"):
        out.append(os.path.join(os.environ["HF_HOME"], "hub"))
    xdg = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    out.append(os.path.join(xdg, "huggingface", "hub"))
    seen, uniq = set(), []
    for d in out:
        d = os.path.abspath(os.path.expanduser(d))
        if d not in seen:
            seen.add(d)
            uniq.append(d)
    return uniq


def _is_pipeline_dir(path: str) -> bool:
    return os.path.isfile(os.path.join(path, "unet", "config.json")) or os.path.isfile(os.path.join(path, "model_index.json"))


def resolve_model_dir(model_id: str) -> Optional[str]:
    """A local diffusers-layout directory for `model_id`: the path itself, or the cached snapshot of a hub id (`org/name`), or None."""
    model_id = str(model_id)
    if os.path.isdir(model_id):
        return model_id
    if model_id.count("/") != 1 or model_id.startswith((".", "/", "~")):
        return None
    folder = "models--" + model_id.replace("/", "--")
    for cache in hub_cache_dirs():
        repo = os.path.join(cache, folder)
        snaps = os.path.join(repo, "snapshots")
        if not os.path.isdir(snaps):
            continue
        cands = []
        ref = os.path.join(repo, "refs", "main")
        if os.path.isfile(ref):
            with open(ref) as f:
                rev = f.read().strip()
            if rev and os.path.isdir(os.path.join(snaps, rev)):
                cands.append(os.path.join(snaps, rev))
        others = sorted((os.path.join(snaps, d) for d in os.listdir(snaps)), key=lambda p: -os.path.getmtime(p))
        cands += [p for p in others if os.path.isdir(p) and p not in cands]
        for p in cands:
            if _is_pipeline_dir(p):
                return p
    return None


def describe_search(model_id: str) -> str:
    folder = "models--" + str(model_id).replace("/", "--")
    return ", ".join(os.path.join(c, folder, "snapshots", "*") for c in hub_cache_dirs())


_STEMS = {"unet": "diffusion_pytorch_model", "vae": "diffusion_pytorch_model", "text_encoder": "model"}


def _load_bin(path: str) -> Dict[str, torch.Tensor]:
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def load_component_state_dict(model_dir: str, component: str) -> Dict[str, torch.Tensor]:
    """State dict of `<model_dir>/<component>/` (or of `model_dir` itself when it IS the component directory).  Order: `<stem>.safetensors`, `<stem>.bin` /
    `pytorch_model.bin`, then the `.fp16` variants in the same order; each candidate as a single file or as its sharded `.index.json`."""
    stem = _STEMS.get(component, "diffusion_pytorch_model")
    dirs = [os.path.join(model_dir, component), model_dir]
    stems = [stem] + (["pytorch_model"] if stem != "pytorch_model" else [])
    tried = []
    for d in dirs:
        if not os.path.isdir(d):
            continue
        # full-precision files of BOTH formats before any `.fp16` variant (a float32 load of a repository that ships `<stem>.bin` next to
        # `<stem>.fp16.safetensors` must not silently get fp16-rounded weights); within a variant every safetensors candidate before any pickle-format .bin
        for variant, ext in (("", ".safetensors"), ("", ".bin"), (".fp16", ".safetensors"), (".fp16", ".bin")):
            for st in stems:
                single = os.path.join(d, st + variant + ext)
                index = single + ".index.json"
                tried.append(single)
                if os.path.isfile(single):
                    if ext == ".safetensors":
                        from safetensors.torch import load_file
                        return load_file(single)
                    return _load_bin(single)
                if os.path.isfile(index):
                    with open(index) as f:
                        wm = json.load(f)["weight_map"]
                    sd: Dict[str, torch.Tensor] = {}
                    for shard in sorted(set(wm.values())):
                        sp = os.path.join(d, shard)
                        if not os.path.isfile(sp):
                            raise FileNotFoundError(f"{index} names the shard {shard}, which is missing")
                        if shard.endswith(".safetensors"):
                            from safetensors.torch import load_file
                            part = load_file(sp)
                        else:
                            part = _load_bin(sp)
                        sd.update(part)
                    lost = [k for k in wm if k not in sd]
                    if lost:
                        raise RuntimeError(f"{index}: {len(lost)} tensors of the weight map are in no shard (first: {lost[:3]})")
                    return sd
    raise FileNotFoundError(f"no weights for '{component}' under {model_dir} (looked for {', '.join(os.path.relpath(t, model_dir) for t in tried[:8])}, ...)")


# what the own modules implement of diffusers' UNet2DConditionModel config space: everything else is refused BY NAME instead of loading into a
# module that computes something else
_UNET_REQUIRED = {
    "layers_per_block": 2, "norm_num_groups": 32, "act_fn": "silu", "center_input_sample": False, "flip_sin_to_cos": True, "freq_shift": 0,
    "mid_block_type": "UNetMidBlock2DCrossAttn", "dual_cross_attention": False, "class_embed_type": None, "addition_embed_type": None,
    "num_class_embeds": None, "time_embedding_type": "positional", "resnet_time_scale_shift": "default", "encoder_hid_dim": None,
    "encoder_hid_dim_type": None, "conv_in_kernel": 3, "conv_out_kernel": 3, "only_cross_attention": False, "transformer_layers_per_block": 1,
    "downsample_padding": 1, "mid_block_scale_factor": 1, "timestep_post_act": None, "time_cond_proj_dim": None, "class_embeddings_concat": False,
    "attention_type": "default", "resnet_skip_time_act": False, "resnet_out_scale_factor": 1.0, "time_embedding_act_fn": None,
    "cross_attention_norm": None, "addition_time_embed_dim": None, "projection_class_embeddings_input_dim": None, "dropout": 0.0,
    "mid_block_only_cross_attention": None, "reverse_transformer_layers_per_block": None, "num_attention_heads": None,
}


def validate_unet_config(cfg: dict) -> None:
    """Raise ValueError naming the first config entry the own UNet does not implement.  Accepted without effect: `upcast_attention` (the attention
    kernel always forms scores and softmax in fp32 -- it IS the upcast form), `use_linear_projection` (SD 1.x stores proj_in / proj_out as 1x1
    convolutions: the same linear map, squeezed at load time), `sample_size`, `norm_eps` at its default, bookkeeping keys starting with '_'."""
    for k, want in _UNET_REQUIRED.items():
        if k in cfg and cfg[k] != want and not (isinstance(want, float) and float(cfg[k]) == want):
            raise ValueError(f"unet/config.json: {k}={cfg[k]!r} is not supported (this build implements {want!r})")
    if float(cfg.get("norm_eps", 1e-5)) != 1e-5:
        raise ValueError(f"unet/config.json: norm_eps={cfg['norm_eps']!r} is not supported (1e-05)")
    boc = list(cfg.get("block_out_channels", (320, 640, 1280, 1280)))
    down = list(cfg.get("down_block_types", ["CrossAttnDownBlock2D"] * (len(boc) - 1) + ["DownBlock2D"]))
    up = list(cfg.get("up_block_types", ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * (len(boc) - 1)))
    if len(down) != len(boc) or len(up) != len(boc):
        raise ValueError("unet/config.json: block type lists and block_out_channels differ in length")
    for t in down:
        if t not in ("CrossAttnDownBlock2D", "DownBlock2D"):
            raise ValueError(f"unet/config.json: down block type {t!r} is not supported")
    for t in up:
        if t not in ("CrossAttnUpBlock2D", "UpBlock2D"):
            raise ValueError(f"unet/config.json: up block type {t!r} is not supported")
    if [t.startswith("CrossAttn") for t in up] != [t.startswith("CrossAttn") for t in reversed(down)]:
        raise ValueError("unet/config.json: up blocks do not mirror the down blocks")
    cad = cfg.get("cross_attention_dim", 1024)
    if isinstance(cad, (list, tuple)):
        raise ValueError("unet/config.json: per-block cross_attention_dim is not supported")


def validate_scheduler_config(cfg: dict) -> None:
    if not cfg:
        return
    if cfg.get("beta_schedule", "scaled_linear") != "scaled_linear":
        raise ValueError(f"scheduler_config.json: beta_schedule {cfg['beta_schedule']!r} is not supported")
    if cfg.get("prediction_type", "epsilon") not in ("epsilon", "v_prediction"):
        raise ValueError(f"scheduler_config.json: prediction_type {cfg['prediction_type']!r} is not supported (epsilon, v_prediction)")
    if cfg.get("timestep_spacing", "leading") != "leading":
        raise ValueError(f"scheduler_config.json: timestep_spacing {cfg['timestep_spacing']!r} is not supported ('leading', the SD configs' value)")
    if cfg.get("thresholding", False) or cfg.get("clip_sample", False):
        raise ValueError("scheduler_config.json: thresholding / clip_sample are not supported (the SD configs switch both off)")
    if cfg.get("rescale_betas_zero_snr", False):
        raise ValueError("scheduler_config.json: rescale_betas_zero_snr is not supported")
