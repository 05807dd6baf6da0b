"""Image <-> latent stage (row X1 of SURVEY.md section 8a; `decode_image` of modified_stable_diffusion_gs.pyc for G1).

Reference: extract.py:39-43 `img_to_latents` = `vae.encode(2x-1).latent_dist.mean * 0.18215` with the stock diffusers
`AutoencoderKL` of stabilityai/stable-diffusion-2-1-base; the bytecode pipelines decode with `vae.decode(latents / 0.18215)`.
diffusers and the weights are not available here, so this is a from-scratch torch module with the published SD VAE shape
(block_out_channels 128/256/512/512, 2 resnets per encoder block, 3 per decoder block, one single-head attention in each
mid block, 32-group GroupNorm eps 1e-6, latent_channels 4, 83.7 M parameters) and diffusers' parameter names so that
`vae/diffusion_pytorch_model.safetensors` loads 1:1.  With synthetic weights it is NOT an autoencoder: it exercises the data
path and its cost, while watermark accuracy through decode->encode is only meaningful with real weights.
"""
from __future__ import annotations

import os
from typing import Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

SCALING_FACTOR = 0.18215  # extract.py:42

# Padded-flat NHWC path (pf.py / csrc/gswm_conv.hip), same scheme as the UNet: every 3x3 / 1x1 / stride-2 convolution is the
# hand-written MFMA implicit GEMM with bias and residual fused in its epilogue, GroupNorm+SiLU the PF kernels; plain torch ops
# remain for CPU / fp32 tensors.
USE_PF = True
FALLBACKS = {}        # (reason -> count) of GPU half-precision calls that ran the plain-torch module path instead of the hand-written kernels


STRICT = __import__("os").environ.get("GSW_STRICT_KERNELS", "1") != "0"      # leaving the hand-written path raises (see unet.STRICT); 0 / --strict_kernels 0: warn and count instead


def _fell_off(what: str, x: torch.Tensor) -> None:
    """A half-precision device tensor is about to run the plain-torch path (MIOpen / hipBLASLt / aotriton): say so, once per reason."""
    if USE_PF and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
        why = f"{what}: input {tuple(x.shape)} {x.dtype} is off the padded-flat path (fp16 / bf16; H, W multiples of 8; mid-block tokens % 8; channel counts multiples of 64)"
        if STRICT:
            raise RuntimeError("gswm vae (strict kernels): " + why + " -- pass --strict_kernels 0 to allow the library kernels")
        if why not in FALLBACKS:
            import warnings
            warnings.warn("gswm vae: " + why + " -- running the plain torch modules", RuntimeWarning, stacklevel=3)
        FALLBACKS[why] = FALLBACKS.get(why, 0) + 1


def _pw(conv: nn.Conv2d, cin_pad: int = 0, cout_pad: int = 0):
    """Packed [N, 9*C] weight (+ bias) of a convolution, optionally zero-padded to `cin_pad` input / `cout_pad` output channels
    (the 3- / 4- / 8-channel edges of the VAE ride on one 64-wide tile)."""
    from .pf import cached, pack_conv_weight

    def build():
        w, b = conv.weight.detach(), conv.bias.detach()
        if cin_pad and w.shape[1] < cin_pad:
            w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - w.shape[1], *w.shape[2:])], dim=1)
        if cout_pad and w.shape[0] < cout_pad:
            w = torch.cat([w, w.new_zeros(cout_pad - w.shape[0], *w.shape[1:])], dim=0)
            b = torch.cat([b, b.new_zeros(cout_pad - b.shape[0])])
        return pack_conv_weight(w), b.contiguous()

    return cached(conv, "_gsw_packed", (conv.weight, conv.bias), build)


def _pw_out_folded(conv_out: nn.Conv2d, post: nn.Conv2d):
    """conv_out followed by a 1x1 convolution (`quant_conv(encoder(x))`, extract.py:41 -> diffusers AutoencoderKL.encode) as ONE 3x3
    convolution: W' = W_post W_out per tap, b' = W_post b_out + b_post (composed in fp32, rounded once), zero-padded to 64 outputs."""
    from .pf import cached, pack_conv_weight

    def build():
        wp = post.weight.detach().float()[:, :, 0, 0]
        w = torch.einsum("oc,cikl->oikl", wp, conv_out.weight.detach().float())
        b = wp @ conv_out.bias.detach().float() + post.bias.detach().float()
        w = torch.cat([w, w.new_zeros(64 - w.shape[0], *w.shape[1:])], dim=0).to(conv_out.weight.dtype)
        b = torch.cat([b, b.new_zeros(64 - b.shape[0])]).to(conv_out.weight.dtype)
        return pack_conv_weight(w), b.contiguous()

    return cached(conv_out, "_gsw_folded", (conv_out.weight, conv_out.bias, post.weight, post.bias), build)


def _pw_in_folded(conv_in: nn.Conv2d, pre: nn.Conv2d):
    """A 1x1 convolution followed by conv_in (`decoder(post_quant_conv(z))`, diffusers AutoencoderKL.decode) as ONE 3x3 convolution over
    [z | 1]: W'[:, c] = W_in W_pre[:, c] per tap, and the 1x1 bias rides on a constant-one input channel (weights W_in b_pre per tap) --
    exact at the image border too, where the zero padding applies to the 1x1 output, not to its input."""
    from .pf import cached, pack_conv_weight

    def build():
        wi = conv_in.weight.detach().float()
        wp, bp = pre.weight.detach().float()[:, :, 0, 0], pre.bias.detach().float()
        w = torch.einsum("oikl,ic->ockl", wi, wp)
        w1 = torch.einsum("oikl,i->okl", wi, bp)[:, None]
        w = torch.cat([w, w1, w.new_zeros(w.shape[0], 64 - w.shape[1] - 1, *w.shape[2:])], dim=1).to(conv_in.weight.dtype)
        return pack_conv_weight(w), conv_in.bias.detach().contiguous()

    return cached(conv_in, "_gsw_folded", (conv_in.weight, conv_in.bias, pre.weight, pre.bias), build)


def _gn_pf(x, norm: nn.GroupNorm, act=True, tokens=False):
    from .pf import groupnorm_pf
    return groupnorm_pf(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act=act, tokens=tokens)


def _pf_ok(x: torch.Tensor) -> bool:
    return USE_PF and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.shape[-2] % 8 == 0


def _convs_fit_pf(module: nn.Module) -> bool:
    """Every convolution but the padded edges (conv_in's input, conv_out's output) needs channel counts in multiples of 64."""
    ok = getattr(module, "_gsw_pf_ok", None)
    if ok is None:
        ok = all((n == "conv_in" or m.in_channels % 64 == 0) and (n == "conv_out" or m.out_channels % 64 == 0)
                 for n, m in module.named_modules() if isinstance(m, nn.Conv2d))
        module._gsw_pf_ok = ok
    return ok


def _to_pf64(x: torch.Tensor, ones_channel: bool = False):
    """NCHW tensor with < 64 channels -> PF tensor with 64 channels (zero-filled; ones_channel: channel C of every real pixel is 1)."""
    from .pf import PF
    B, C, H, W = x.shape
    p = PF.zeros(B, H, W, 64, x.dtype, x.device)
    p.interior[..., :C].copy_(x.permute(0, 2, 3, 1))
    if ones_channel:
        p.interior[..., C].fill_(1.0)
    return p


class VaeResnet(nn.Module):
    def __init__(self, cin, cout, groups=32, eps=1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h

    def forward_pf(self, x):
        from .pf import conv_pf
        h = conv_pf(_gn_pf(x, self.norm1), *_pw(self.conv1))
        h = _gn_pf(h, self.norm2)
        sc = x if self.conv_shortcut is None else conv_pf(x, *_pw(self.conv_shortcut), ksize=1)
        return conv_pf(h, *_pw(self.conv2), resid=sc)                   # residual add in the GEMM epilogue


class VaeAttention(nn.Module):
    def __init__(self, ch, groups=32, eps=1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=eps)
        self.to_q = nn.Linear(ch, ch)
        self.to_k = nn.Linear(ch, ch)
        self.to_v = nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch)])

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(y)[:, None], self.to_k(y)[:, None], self.to_v(y)[:, None]   # one head of width c
        o = F.scaled_dot_product_attention(q, k, v)[:, 0]
        return x + self.to_out[0](o).transpose(1, 2).reshape(b, c, h, w)

    def forward_pf(self, x):
        from .pf import attention_single_head, cached, gemm
        y = _gn_pf(x, self.group_norm, act=False, tokens=True)          # GroupNorm writes dense tokens [B, H*W, C]
        C, S = x.C, x.H * x.W
        if C % 64 or S % 8:          # (Encoder._pf_tokens_ok / _pf_ok keep such inputs on the plain-torch path with a loud warning)
            raise RuntimeError(f"VaeAttention: {C} channels x {S} tokens is off the hand-written path (channels % 64, tokens % 8)")
        # q | k from one GEMM over the tokens, V^T from a transposing GEMM, softmax(q k^T) v per image on the matmul engine, and the output
        # projection + residual written straight into the PF tensor's interior rows
        wqk, bqk = cached(self, "_gsw_wqk", (self.to_q.weight, self.to_k.weight, self.to_q.bias, self.to_k.bias),
                          lambda: (torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach()]).contiguous(),
                                   torch.cat([self.to_q.bias.detach(), self.to_k.bias.detach()]).contiguous()))
        qk = gemm(y, wqk, bqk)
        vt = gemm(y, self.to_v.weight.detach(), self.to_v.bias.detach(), mode="trans", tokens=S)
        o = attention_single_head(qk[..., :C], qk[..., C:], vt)
        gemm(o, self.to_out[0].weight.detach(), self.to_out[0].bias.detach(), resid=x.rows, mode="tok2pf", tokens=S, width=x.W, out=x.rows, stats_for=x)
        return x


class VaeMid(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(ch, ch), VaeResnet(ch, ch)])
        self.attentions = nn.ModuleList([VaeAttention(ch)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))

    def forward_pf(self, x):
        return self.resnets[1].forward_pf(self.attentions[0].forward_pf(self.resnets[0].forward_pf(x)))


class _Down(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))     # diffusers' asymmetric padding for the VAE downsampler

    def forward_pf(self, x):
        from .pf import conv_pf
        return conv_pf(x, *_pw(self.conv), stride=2, pad_after_only=True)


class _Up(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))

    def forward_pf(self, x):
        from .pf import PF, cached, conv_pf, conv_up2x_fusable, conv_up2x_pf, pack_upsample_weight
        if conv_up2x_fusable(x, self.conv.out_channels):
            # sub-pixel form (gsw_conv_up2x_pf, as in the UNet): four 2x2 convolutions of the LOW-resolution tensor -- 2.25x fewer FLOPs, no upsampled
            # intermediate, no copy kernels.  The three upsamplers are 56 % of the decoder's FLOPs in the 3x3-on-upsampled form.
            w4 = cached(self, "_gsw_up4", (self.conv.weight,), lambda: pack_upsample_weight(self.conv.weight))
            return conv_up2x_pf(x, w4, self.conv.bias)
        up = PF.zeros(x.B, 2 * x.H, 2 * x.W, x.C, x.buf.dtype, x.buf.device)
        xi, g = x.interior, up.grid
        for dy in (0, 1):
            for dx in (0, 1):
                g[:, 1 + dy:1 + dy + 2 * x.H:2, 1 + dx:1 + dx + 2 * x.W:2, :].copy_(xi)
        return conv_pf(up, *_pw(self.conv))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, down):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin, cout), VaeResnet(cout, cout)])
        self.downsamplers = nn.ModuleList([_Down(cout)]) if down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)

    def forward_pf(self, x):
        for r in self.resnets:
            x = r.forward_pf(x)
        return x if self.downsamplers is None else self.downsamplers[0].forward_pf(x)


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, up):
        super().__init__()
        self.resnets = nn.ModuleList([VaeResnet(cin if i == 0 else cout, cout) for i in range(3)])
        self.upsamplers = nn.ModuleList([_Up(cout)]) if up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)

    def forward_pf(self, x):
        for r in self.resnets:
            x = r.forward_pf(x)
        return x if self.upsamplers is None else self.upsamplers[0].forward_pf(x)


class Encoder(nn.Module):
    def __init__(self, chs: Sequence[int], latent=4):
        super().__init__()
        self.conv_in = nn.Conv2d(3, chs[0], 3, padding=1)
        self.down_blocks = nn.ModuleList([_EncBlock(chs[max(i - 1, 0)], c, i < len(chs) - 1) for i, c in enumerate(chs)])
        self.mid_block = VaeMid(chs[-1])
        self.conv_norm_out = nn.GroupNorm(32, chs[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(chs[-1], 2 * latent, 3, padding=1)

    def forward(self, x, post: nn.Conv2d = None):
        """post: the 1x1 `quant_conv` that follows the encoder -- folded into conv_out on the hand-written path"""
        if _pf_ok(x) and self._pf_shapes_ok() and self._pf_tokens_ok(x):
            return self.forward_pf(x, post)
        _fell_off("Encoder", x)
        x = self.conv_in(x)
        for b in self.down_blocks:
            x = b(x)
        x = self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))
        return x if post is None else post(x)

    def _pf_shapes_ok(self):
        return _convs_fit_pf(self)

    def _pf_tokens_ok(self, x):
        """the mid block's single-head attention takes token counts in multiples of 8"""
        f = 1 << (len(self.down_blocks) - 1)
        return ((x.shape[-2] // f) * (x.shape[-1] // f)) % 8 == 0

    def forward_pf(self, x, post: nn.Conv2d = None):
        from .pf import conv_pf
        h = conv_pf(_to_pf64(x), *_pw(self.conv_in, cin_pad=64))
        for b in self.down_blocks:
            h = b.forward_pf(h)
        h = _gn_pf(self.mid_block.forward_pf(h), self.conv_norm_out)
        y = conv_pf(h, *(_pw(self.conv_out, cout_pad=64) if post is None else _pw_out_folded(self.conv_out, post)))
        return y.interior[..., : self.conv_out.out_channels].permute(0, 3, 1, 2).contiguous()


class Decoder(nn.Module):
    def __init__(self, chs: Sequence[int], latent=4):
        super().__init__()
        rev = list(reversed(chs))
        self.conv_in = nn.Conv2d(latent, rev[0], 3, padding=1)
        self.mid_block = VaeMid(rev[0])
        self.up_blocks = nn.ModuleList([_DecBlock(rev[max(i - 1, 0)], c, i < len(rev) - 1) for i, c in enumerate(rev)])
        self.conv_norm_out = nn.GroupNorm(32, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], 3, 3, padding=1)

    def forward(self, z, pre: nn.Conv2d = None):
        """pre: the 1x1 `post_quant_conv` that precedes the decoder -- folded into conv_in on the hand-written path"""
        if _pf_ok(z) and self._pf_shapes_ok():
            return self.forward_pf(z, pre)
        _fell_off("Decoder", z)
        x = self.mid_block(self.conv_in(z if pre is None else pre(z)))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))

    def _pf_shapes_ok(self):
        return _convs_fit_pf(self)

    def forward_pf(self, z, pre: nn.Conv2d = None):
        from .pf import conv_pf
        h0 = conv_pf(_to_pf64(z), *_pw(self.conv_in, cin_pad=64)) if pre is None else conv_pf(_to_pf64(z, ones_channel=True), *_pw_in_folded(self.conv_in, pre))
        h = self.mid_block.forward_pf(h0)
        for b in self.up_blocks:
            h = b.forward_pf(h)
        y = conv_pf(_gn_pf(h, self.conv_norm_out), *_pw(self.conv_out, cout_pad=64))
        return y.interior[..., : self.conv_out.out_channels].permute(0, 3, 1, 2).contiguous()


class AutoencoderKL(nn.Module):
    def __init__(self, block_out_channels: Sequence[int] = (128, 256, 512, 512), latent_channels: int = 4):
        super().__init__()
        self.encoder = Encoder(block_out_channels, latent_channels)
        self.decoder = Decoder(block_out_channels, latent_channels)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.latent_channels = latent_channels

    def encode_mean(self, x: torch.Tensor) -> torch.Tensor:
        """`vae.encode(x).latent_dist.mean` (extract.py:41-42): first half of the moments."""
        return self.encoder(x, post=self.quant_conv)[:, : self.latent_channels]

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        return self.decoder(z, pre=self.post_quant_conv)


@torch.no_grad()
def img_to_latents(x: torch.Tensor, vae: AutoencoderKL) -> torch.Tensor:
    """extract.py:39-43: x in [0,1] -> 2x-1 -> posterior mean * 0.18215."""
    return (vae.encode_mean(2.0 * x - 1.0) * SCALING_FACTOR).contiguous()


@torch.no_grad()
def normalised_img_to_latents(xn: torch.Tensor, vae: AutoencoderKL) -> torch.Tensor:
    """img_to_latents for an input that already is 2x-1 (imaging.resize_lanczos / jpeg_roundtrip with out='f16' fuse it)."""
    return (vae.encode_mean(xn) * SCALING_FACTOR).contiguous()


@torch.no_grad()
def latents_to_img(latents: torch.Tensor, vae: AutoencoderKL) -> torch.Tensor:
    """`decode_image` + `torch_to_numpy` prefix of the bytecode pipelines: vae.decode(latents / 0.18215) -> (x/2+0.5).clamp(0,1)."""
    return (vae.decode(latents / SCALING_FACTOR) / 2 + 0.5).clamp(0, 1)


def synthetic_init_(model: nn.Module, seed: int = 0) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                std = (1.0 / p[0].numel()) ** 0.5
                if name.endswith(("conv2.weight", "to_out.0.weight")):
                    std *= 0.3
                p.copy_(torch.randn(p.shape, generator=g) * std)
            else:
                p.fill_(1.0 if name.endswith("weight") and "norm" in name else 0.0)
    return model


def load_diffusers_state_dict(model: nn.Module, weight_dir: str) -> nn.Module:
    from .checkpoint import load_component_state_dict
    sd = load_component_state_dict(weight_dir, "vae")          # safetensors, sharded safetensors or .bin
    # older diffusers checkpoints name the mid-block attention projections query/key/value/proj_attn
    ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
    fixed = {}
    for k, v in sd.items():
        for old, new in ren.items():
            k = k.replace(f"attentions.0.{old}.", f"attentions.0.{new}.")
        if k.endswith(("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight")) and v.dim() == 4:
            v = v[:, :, 0, 0]
        fixed[k] = v
    missing, unexpected = model.load_state_dict(fixed, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"VAE state dict mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}")
    return model
