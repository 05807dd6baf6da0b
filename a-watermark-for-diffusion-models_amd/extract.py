"""Drop-in twin of the recover-bits API of the reference's extract.py (extract.py:72-110), with the per-element
norm.cdf loop, the ChaCha20 decrypt and the majority vote running as one HIP kernel.

`args` is the reference's argparse namespace: .key (32 bytes), .nonce (16 bytes), .l (must be 1: the reference's
l > 1 path is non-functional), .message_length.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native as N
from . import codec

_TORCH_OK = (torch.float16, torch.bfloat16, torch.float32, torch.float64)


def _to_device_latents(reversed_latents, device):
    """np.nditer order (extract.py:82): memory order of the array, i.e. C order for the contiguous tensors the
    inversion returns.  One image per call, any shape."""
    if isinstance(reversed_latents, torch.Tensor):
        t = reversed_latents.detach()
        if t.dtype not in _TORCH_OK:
            t = t.to(torch.float64)
        return t.contiguous().to(device).reshape(1, -1)
    a = np.asarray(reversed_latents)
    if a.dtype not in (np.float16, np.float32, np.float64):
        a = a.astype(np.float64)
    a = np.ascontiguousarray(a.ravel(order="K"))
    return torch.from_numpy(a).to(device).reshape(1, -1)


def recover_exactracted_message(reversed_latents, args, *, device="cuda"):
    """extract.py:72-101 -> str of message_length '0'/'1' characters.

    Raises ValueError where the reference does (a latent >= 8.2924 saturates norm.cdf so int(y) == 2, or NaN;
    extract.py:84-86) and IndexError when the padded bit count is not a multiple of message_length (extract.py:98).
    """
    if int(getattr(args, "l", 1)) != 1:
        raise ValueError("only l == 1 is functional in the reference (extract.py:84-86 breaks for l > 1)")
    z = _to_device_latents(reversed_latents, device)
    m = int(args.message_length)
    bits, flags = codec.extract_batch(z, args.key, args.nonce, m)
    f = int(flags[0].item())
    if f & N.GSW_FLAG_NAN:
        raise ValueError("cannot convert float NaN to integer")
    if f & N.GSW_FLAG_SATURATED:
        raise ValueError("invalid literal for int() with base 2")
    return codec.bits_to_str(bits[0].cpu().numpy())[:m]


def recover_exactracted_message_batch(latents: torch.Tensor, args):
    """Batch form: latents [B, 4, h, w] on the device -> (list of bit strings or the raised exception per image).
    Mirrors the per-image try/except of extract.py:148-155."""
    m = int(args.message_length)
    bits, flags = codec.extract_batch(latents.contiguous(), args.key, args.nonce, m)
    bits_h, flags_h = bits.cpu().numpy(), flags.cpu().numpy()
    out = []
    for b in range(bits_h.shape[0]):
        if flags_h[b] & N.GSW_FLAG_NAN:
            out.append(ValueError("cannot convert float NaN to integer"))
        elif flags_h[b] & N.GSW_FLAG_SATURATED:
            out.append(ValueError("invalid literal for int() with base 2"))
        else:
            out.append(codec.bits_to_str(bits_h[b])[:m])
    return out


def calculate_bit_accuracy(original_message_hex, extracted_message_bin):
    """extract.py:103-110 (host string arithmetic; the batched device form is codec.bit_matches)."""
    original_message_bin = bin(int(original_message_hex, 16))[2:].zfill(len(original_message_hex) * 4)
    min_length = min(len(original_message_bin), len(extracted_message_bin))
    original_message_bin = original_message_bin[:min_length]
    extracted_message_bin = extracted_message_bin[:min_length]
    matching_bits = sum(1 for x, y in zip(original_message_bin, extracted_message_bin) if x == y)
    return original_message_bin, matching_bits / min_length


# =====================================================================================================================
# X1 + X2 + H1: the rest of extract.py -- image -> latents -> DDIM inversion, the per-image / per-directory harness and
# the CLI.  Same function names, `args` fields, stdout / result.txt text as the reference (extract.py:23-70,112-211).
# Differences that are the point of the port: the models are loaded ONCE (the reference calls from_pretrained per image,
# extract.py:56-60), images of a directory go through the UNet loop as a batch, and the inverted latents stay on the
# device for the vote (the reference returns `.cpu()`, extract.py:70).
# =====================================================================================================================
import glob
import os
import sys
from datetime import datetime

_MODEL_CACHE = {}


class Models:
    """UNet + VAE (+ the context of the empty prompt) of one `model_id`.

    `model_id` is a local directory in diffusers layout (unet/, vae/, optionally text_encoder/ + tokenizer/): the weights
    are loaded from it.  Anything else (e.g. the reference's default hub id 'stabilityai/stable-diffusion-2-1-base') cannot
    be downloaded here, so seeded synthetic weights of the same architecture are used and a warning is printed -- the data
    path and its cost are real, the recovered bits are only meaningful with real weights."""

    def __init__(self, model_id, device="cuda", dtype=torch.float16):
        from . import unet as U, vae as V
        self.device, self.dtype = torch.device(device), dtype
        self.synthetic = not os.path.isdir(str(model_id))
        self.unet = U.UNet2DCondition()
        self.vae = V.AutoencoderKL()
        if self.synthetic:
            print(f"[gswm] '{model_id}' is not a local diffusers directory: using synthetic weights (results are not meaningful)", file=sys.stderr)
            U.synthetic_init_(self.unet, 0)
            V.synthetic_init_(self.vae, 1)
        else:
            U.load_diffusers_state_dict(self.unet, model_id)
            V.load_diffusers_state_dict(self.vae, model_id)
        self.unet.to(self.device, dtype).eval()
        self.vae.to(self.device, dtype).eval()
        self.prediction_type = "epsilon"
        cfg = os.path.join(str(model_id), "scheduler", "scheduler_config.json")
        if os.path.exists(cfg):
            import json
            self.prediction_type = json.load(open(cfg)).get("prediction_type", "epsilon")
        self.ctx_empty = self._empty_prompt_context(model_id)

    def _empty_prompt_context(self, model_id):
        """Context of prompt "" (extract.py:66): CLIP text encoder when the directory ships one, else a fixed synthetic tensor."""
        te = os.path.join(str(model_id), "text_encoder")
        if os.path.isdir(te):
            from transformers import CLIPTextModel, CLIPTokenizer
            tok = CLIPTokenizer.from_pretrained(os.path.join(str(model_id), "tokenizer"))
            enc = CLIPTextModel.from_pretrained(te).to(self.device, self.dtype).eval()
            ids = tok("", padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids.to(self.device)
            with torch.no_grad():
                return enc(ids)[0].to(self.dtype)
        g = torch.Generator().manual_seed(77)
        return torch.randn(1, 77, 1024, generator=g).to(self.device, self.dtype)


def load_models(model_id, device="cuda", dtype=torch.float16) -> Models:
    k = (str(model_id), str(device), dtype)
    if k not in _MODEL_CACHE:
        _MODEL_CACHE[k] = Models(model_id, device, dtype)
    return _MODEL_CACHE[k]


def load_image(imgname, target_size=None) -> torch.Tensor:
    """extract.py:31-37: PIL RGB, optional Lanczos resize to (w, h), ToTensor -> float32 [1,3,H,W] in [0,1]."""
    from PIL import Image
    pil_img = Image.open(imgname).convert("RGB")
    if target_size is not None:
        if isinstance(target_size, int):
            target_size = (target_size, target_size)
        pil_img = pil_img.resize(tuple(target_size), Image.Resampling.LANCZOS)
    a = np.asarray(pil_img, dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.0)[None]


def load_images_device(image_paths, target_size=None, *, device="cuda", out="f16") -> torch.Tensor:
    """Batch form of load_image + the `.to(float16)` / `2.*x - 1.` that follow it (extract.py:31-37,48,40), resized and normalised on
    the device (imaging.resize_lanczos, bit-identical to the PIL / torchvision chain).  File decoding stays on the host; images of
    one call must share a size (they do in the reference's use: all outputs of one generation run)."""
    from PIL import Image
    from . import imaging
    arrs = [np.asarray(Image.open(p).convert("RGB"), dtype=np.uint8) for p in image_paths]
    if isinstance(target_size, int):
        target_size = (target_size, target_size)
    groups = {}
    for i, a in enumerate(arrs):
        groups.setdefault(a.shape, []).append(i)
    outs = [None] * len(arrs)
    for shape, idx in groups.items():
        dev = torch.from_numpy(np.stack([arrs[i] for i in idx])).to(device)
        res = imaging.resize_lanczos(dev, None if target_size is None else tuple(target_size), out=out)
        for j, i in enumerate(idx):
            outs[i] = res[j]
    if len({tuple(o.shape) for o in outs}) != 1:
        raise ValueError("images of one batch must have one size after resizing (pass target_size)")
    return torch.stack(outs)


def img_to_latents(x: torch.Tensor, vae):
    """extract.py:39-43."""
    from . import vae as V
    return V.img_to_latents(x, vae)


def _scheduler_steps(args, models):
    from .ddim import DDIMSchedule
    if args.scheduler == "DDIM":
        return DDIMSchedule(num_inference_steps=int(args.num_inference_steps), prediction_type=models.prediction_type)
    if args.scheduler == "DPMs":
        from .ddim import DPMSolverInverseSchedule
        return DPMSolverInverseSchedule(num_inference_steps=int(args.num_inference_steps), prediction_type=models.prediction_type)
    raise ValueError("Please choose 'DPMs' or 'DDIM' for the scheduler.")     # extract.py:54


@torch.no_grad()
def exactract_latents_batch(image_paths, args, *, device="cuda") -> torch.Tensor:
    """Batch form of exactract_latents: [B,4,h,w] fp16 latents on the DEVICE."""
    from .ddim import ddim_invert, dpms_invert, DPMSolverInverseSchedule
    models = load_models(args.model_id, device)
    sched = _scheduler_steps(args, models)
    from . import vae as V
    if models.dtype == torch.float16:
        xn = load_images_device(image_paths, [args.width, args.height], device=models.device, out="f16")      # = 2 * fp16(ToTensor) - 1
    else:
        xn = 2.0 * load_images_device(image_paths, [args.width, args.height], device=models.device, out="f32").to(models.dtype) - 1.0
    latents = V.normalised_img_to_latents(xn, models.vae)
    ctx = models.ctx_empty.expand(latents.shape[0], -1, -1)
    if isinstance(sched, DPMSolverInverseSchedule):
        return dpms_invert(models.unet, latents, ctx, sched)
    return ddim_invert(models.unet, latents, ctx, sched)


def exactract_latents(args, *, device="cuda") -> torch.Tensor:
    """extract.py:46-70 -> fp16 CPU tensor [1,4,h,w] like the reference (use exactract_latents_batch to stay on the device)."""
    return exactract_latents_batch([args.single_image_path], args, device=device).cpu()


def get_result_for_one_image(args):
    """extract.py:112-117 (same stdout text)."""
    reversed_latents = exactract_latents_batch([args.single_image_path], args)
    extracted_message_bin = recover_exactracted_message(reversed_latents, args)
    original_message_bin, bit_accuracy = calculate_bit_accuracy(args.original_message_hex, extracted_message_bin)
    print(f"{os.path.basename(args.single_image_path)}\nOriginal Message: {original_message_bin} \nExtracted Message: {extracted_message_bin}\nBit Accuracy: {bit_accuracy}\n")
    return original_message_bin, extracted_message_bin, bit_accuracy


def write_batch_info(result_file, args):
    """extract.py:165-175."""
    result_file.write("=" * 40 + "Batch Info" + "=" * 40 + "\n")
    current_time = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
    result_file.write(f"Time,{str(current_time)}\n")
    result_file.write(f"key_hex,{args.key_hex}\n")
    result_file.write(f"nonce_hex,{args.nonce_hex}\n")
    result_file.write(f"original_message_hex,{args.original_message_hex}\n")
    result_file.write(f"num_inference_steps,{args.num_inference_steps}\n")
    result_file.write(f"scheduler,{args.scheduler}\n")
    result_file.write("=" * 40 + "Batch Start" + "=" * 40 + "\n")


def process_single_directory(dir_path, args, *, batch_size=16):
    """extract.py:134-163: same result.txt lines; images go through the inversion loop `batch_size` at a time."""
    image_files = glob.glob(os.path.join(dir_path, "*.png")) + glob.glob(os.path.join(dir_path, "*.jpg"))
    if not image_files:
        return
    total_bit_accuracy = 0
    processed_images = 0
    result_file_path = os.path.join(dir_path, "result.txt")
    with open(result_file_path, "a") as result_file:
        write_batch_info(result_file, args)
        for i in range(0, len(image_files), batch_size):
            chunk = image_files[i:i + batch_size]
            try:
                latents = exactract_latents_batch(chunk, args)
                results = recover_exactracted_message_batch(latents, args)
            except Exception as e:  # a failure of the whole chunk (unreadable image, ...) is reported per image like the reference
                results = [e] * len(chunk)
            for image_path, res in zip(chunk, results):
                if isinstance(res, Exception):
                    print(f"Error processing {image_path}: {res}\n")
                    result_file.write(f"Error processing {image_path}: {res}\n")
                    continue
                original_message_bin, bit_accuracy = calculate_bit_accuracy(args.original_message_hex, res)
                print(f"{os.path.basename(image_path)}\nOriginal Message: {original_message_bin} \nExtracted Message: {res}\nBit Accuracy: {bit_accuracy}\n")
                result_file.write(f"{os.path.basename(image_path)}, Bit Accuracy, {bit_accuracy}\n")
                total_bit_accuracy += float(bit_accuracy)
                processed_images += 1
        if processed_images > 0:
            average_bit_accuracy = total_bit_accuracy / processed_images
            result_file.write(f"Average Bit Accuracy, {average_bit_accuracy}\n\n")
            result_file.write("=" * 40 + "Batch End" + "=" * 40 + "\n")
            parent_dir = os.path.dirname(dir_path)
            with open(os.path.join(parent_dir, "result.txt"), "a") as parent_result_file:
                parent_result_file.write(f"{os.path.basename(dir_path)}, Average Bit Accuracy, {average_bit_accuracy}\n")


def process_directory(args):
    """extract.py:120-132."""
    if int(args.is_traverse_subdirectories) == 1:
        with open(os.path.join(args.images_directory_path, "result.txt"), "a") as root_result_file:
            write_batch_info(root_result_file, args)
        for root, dirs, files in os.walk(args.images_directory_path):
            print("=" * 20 + root + "=" * 20)
            for d in dirs:
                process_single_directory(os.path.join(root, d), args)
        with open(os.path.join(args.images_directory_path, "result.txt"), "a") as root_result_file:
            root_result_file.write("=" * 40 + "Batch End" + "=" * 40 + "\n\n")
    else:
        process_single_directory(args.images_directory_path, args)


def build_parser():
    """extract.py:180-195: same flags and defaults."""
    import argparse
    parser = argparse.ArgumentParser(description="Extract watermark from a image")
    parser.add_argument("--model_id", default="stabilityai/stable-diffusion-2-1-base")
    parser.add_argument("--images_directory_path", default="", help="The path of directory containing images to process")
    parser.add_argument("--single_image_path", default="")
    parser.add_argument("--key_hex", required=True, help="Hexadecimal key used for encryption")
    parser.add_argument("--nonce_hex", required=True, help="Hexadecimal nonce used for encryption, It will use the fixed part of the key if nonce is none")
    parser.add_argument("--original_message_hex", required=True, help="Hexadecimal representation of the original message for accuracy calculation")
    parser.add_argument("--num_inference_steps", default=30, type=int, help="Number of inference steps for the model")
    parser.add_argument("--scheduler", default="DDIM", help="Choose a scheduler between 'DPMs' and 'DDIM' to inverse the image")
    parser.add_argument("--is_traverse_subdirectories", default=0, help="Whether to traverse subdirectories recursively")
    parser.add_argument("--l", default=1, type=int, help="The size of slide windows for m")
    parser.add_argument("--width", type=int, default=1024, help="Width of the input image")
    parser.add_argument("--height", type=int, default=1024, help="Height of the input image")
    parser.add_argument("--message_length", type=int, default=1024, help="Length of the message in bits")
    return parser


def main(argv=None):
    """extract.py:179-211."""
    args = build_parser().parse_args(argv)
    args.key = bytes.fromhex(args.key_hex)
    if args.nonce_hex != "":
        args.nonce = bytes.fromhex(args.nonce_hex)
    else:
        args.nonce = bytes.fromhex(args.key_hex[16:48])
    if args.images_directory_path != "":
        process_directory(args)
    elif args.single_image_path != "":
        get_result_for_one_image(args)
    else:
        print("Please set the argument 'images_directory_path' or 'single_image_path'")


if __name__ == "__main__":
    main()
