"""eps-model of the DDIM loops (rows X2 / G1 of SURVEY.md section 8a): a from-scratch Stable-Diffusion-2.1-base-shaped
conditional UNet in plain torch, used as the `eps_model(latents, t, ctx)` callable of ddim.py.

Why it exists: the reference delegates this to diffusers==0.26.0 `UNet2DConditionModel` + HF weights
`stabilityai/stable-diffusion-2-1-base` (extract.py:56-60,66-69), neither of which is available here (no network).  The
architecture below follows the published SD 2.1-base `unet/config.json` (block_out_channels 320/640/1280/1280,
2 resnets per block, 1 transformer layer per attention block, heads 5/10/20/20 x 64, cross_attention_dim 1024, linear
projections, GEGLU feed-forward, 32-group GroupNorm) and keeps diffusers' parameter names, so a diffusers-layout
safetensors state dict loads 1:1 (`load_diffusers_state_dict`).  Without a weight directory the weights are seeded
synthetic -- throughput is representative, image content is not.

On the GPU (fp16 / bf16) every convolution, linear layer and attention runs on the hand-written MFMA kernels of libgswm (PF implicit-GEMM
convolutions, the gsw_gemm matmul engine, the flash-attention kernel); the plain torch ops remain for CPU / meta tensors (FLOP counting,
CPU baseline) and are counted in `FALLBACKS` when a GPU call has to use them.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# Fused HIP elementwise kernels (libgswm: gsw_groupnorm_silu, gsw_geglu) replace torch's GroupNorm -> SiLU (-> broadcast add)
# and GELU -> mul chains on the GPU; the plain torch ops remain for CPU / meta tensors (FLOP counting, CPU baseline).
FUSED_KERNELS = True


def _fusable(x: torch.Tensor) -> bool:
    return FUSED_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16, torch.float32) and x.is_contiguous() \
        and (x.shape[-1] * x.shape[-2]) % 8 == 0


def gn_act(x: torch.Tensor, norm: nn.GroupNorm, act: bool = True, pre_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(GroupNorm(x + pre_bias[:, :, None, None]))"""
    if _fusable(x):
        from . import codec
        return codec.groupnorm_silu(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act=act, pre_bias=pre_bias)
    if pre_bias is not None:
        x = x + pre_bias[:, :, None, None]
    y = norm(x)
    return F.silu(y) if act else y


# Padded-flat NHWC path: every 3x3 / 1x1 / stride-2 convolution of the UNet runs as a hand-written MFMA implicit GEMM (the matmul engine of
# csrc/gswm_mm.hip through csrc/gswm_conv.hip; conv_in / conv_out's 4-channel side padded to one 64-wide tile) with bias, time-embedding and
# residual adds fused into its epilogue, GroupNorm+SiLU as the PF kernels, Upsample2D as four sub-pixel launches; the transformer blocks run on
# the same engine (OWN_GEMM) and the flash-attention kernel (OWN_ATTENTION).
USE_PF = True


def _pw(conv: nn.Conv2d) -> torch.Tensor:
    from .pf import cached, pack_conv_weight
    return cached(conv, "_gsw_packed", (conv.weight,), lambda: pack_conv_weight(conv.weight.detach()))


UPSAMPLE_SUBPIXEL = True   # Upsample2D = four 2x2 convolutions of the low-resolution tensor (gsw_conv_up2x_pf) instead of upsample + 3x3

CACHE_CONTEXT_KV = True   # cross-attention K / V^T of a context tensor are computed once and reused across the steps of a loop

FUSED_QK = True       # self-attention: q and k projections as one GEMM (own attention kernel reads them as column slices)

FUSED_QKV = False     # ... and the transposed value projection from the same launch (gsw_gemm_qkv: the tokens are read once).  Measured: no gain --
                      # forward 113.6 vs 113.3 ms at 128 rows, 60.5 vs 60.2 at 64 (the value part loses the 12-wave form, the launch runs at 235 registers);
                      # kept as an A/B switch, bit-identical to the two launches (tests/test_gpu_qkv.py)

OWN_ATTENTION = True  # attention with head_dim 40 / 64 / 80 / 160, any query count, keys % 8 == 0 runs on gsw_attention instead of torch SDPA

OWN_GEMM = True       # every dense linear layer (q / k / v / out projections, proj_in / proj_out, feed-forward, time embedding) on the
                      # hand-written matmul engine (gsw_gemm, csrc/gswm_mm.hip) with bias / residual / GEGLU / V^T epilogues fused in

FALLBACKS = {}        # (reason -> count) of GPU half-precision calls that left the hand-written path; bench and the full-size tests assert it stays empty

# Leaving the hand-written path (a shape off the engine's K % 64 / N % 8 grid, attention keys % 8, a lattice the padded-flat layout does not take) RAISES by default: a
# 200 x 136 image must not silently run a different backend (hipBLASLt / MIOpen / aotriton).  GSW_STRICT_KERNELS=0, `--strict_kernels 0` of the harness or
# `unet.STRICT = vae.STRICT = False` opt into the library kernels; every such call is then counted in FALLBACKS and warned about once per reason.
STRICT = __import__("os").environ.get("GSW_STRICT_KERNELS", "1") != "0"


def _note_fallback(why: str):
    if STRICT:
        raise RuntimeError("gswm unet (strict kernels): " + why + " -- this call would run a library kernel (hipBLASLt / MIOpen / SDPA); "
                           "pass --strict_kernels 0 to allow it")
    if why not in FALLBACKS:       # loud, once per reason: a GPU run that leaves the hand-written kernels should never be silent
        import warnings
        warnings.warn("gswm unet: " + why, RuntimeWarning, stacklevel=3)
    FALLBACKS[why] = FALLBACKS.get(why, 0) + 1


def _own_gemm_ok(x: torch.Tensor, K: int, N: int) -> bool:
    """The matmul engine takes fp16 / bf16 device tensors with K % 64 == 0 and N % 8 == 0 (every SD 1.x / 2.x linear)."""
    if not (OWN_GEMM and USE_PF and FUSED_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    if K % 64 or N % 8:
        _note_fallback(f"linear K={K} N={N}: library GEMM")
        return False
    return True


def _wb(lin: nn.Linear, x: torch.Tensor):
    """(weight, bias) of a linear layer as the engine wants them: contiguous, the activations' dtype (a fp32 parameter under an fp16
    activation must not be read as fp16 bytes)."""
    if lin.weight.dtype != x.dtype or lin.weight.device != x.device or not lin.weight.is_contiguous():
        raise ValueError(f"linear weight is {lin.weight.dtype} on {lin.weight.device}, activations are {x.dtype} on {x.device}")
    return lin.weight, lin.bias


def _lin(x: torch.Tensor, lin: nn.Linear, resid: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, rowstats: bool = False) -> torch.Tensor:
    """Linear (+ residual in the epilogue) on the matmul engine, else torch.  out: written in place (same shape, contiguous).
    rowstats: the launch also leaves the row records a following LayerNorm can be folded from (pf.ln_stat) on the result."""
    if _own_gemm_ok(x, lin.in_features, lin.out_features):
        from .pf import gemm
        w, b = _wb(lin, x)
        return gemm(x.contiguous(), w, b, resid=None if resid is None else resid.contiguous(), out=out, rowstats=rowstats)
    y = lin(x)
    y = y if resid is None else y + resid
    return y if out is None else out.copy_(y)


def _lin_t(x: torch.Tensor, lin: nn.Linear, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B, S, K] -> (lin(x))^T = [B, N, S]: the value projection in the layout the attention kernel consumes."""
    b, n, _ = x.shape
    if _own_gemm_ok(x, lin.in_features, lin.out_features) and n % 8 == 0:
        from .pf import gemm
        w, bias = _wb(lin, x)
        return gemm(x.contiguous(), w, bias, mode="trans", tokens=n, out=out)
    vt = torch.bmm(lin.weight.unsqueeze(0).expand(b, -1, -1), x.transpose(1, 2))
    vt = vt if lin.bias is None else vt + lin.bias[None, :, None]
    return vt if out is None else out.copy_(vt)


def _gn_pf(x, norm: nn.GroupNorm, act=True, tokens=False):
    from .pf import groupnorm_pf
    return groupnorm_pf(x, norm.weight, norm.bias, norm.num_groups, norm.eps, act=act, tokens=tokens)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """Sinusoidal embedding, flip_sin_to_cos=True, freq_shift=0 (SD config): [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return _lin(F.silu(_lin(x, self.linear_1)), self.linear_2)


class TembRows:
    """silu(time embedding) [B, temb] and, per resnet, its time_emb_proj output as a column slice [B, cout] of ONE GEMM over the concatenated
    projection weights (22 resnets -> one launch instead of 22; the convolution epilogue reads the slice through its row stride)."""
    __slots__ = ("act", "rows", "table")

    def __init__(self, act, rows, table=None):
        self.act, self.rows, self.table = act, rows, table      # table: the [B, sum of channels] matrix the row slices view


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_dim, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_dim, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb_act):
        h = self.conv1(gn_act(x, self.norm1))
        h = self.conv2(gn_act(h, self.norm2, pre_bias=self.time_emb_proj(temb_act)))     # temb add folded into the norm kernel
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h

    def forward_pf(self, x, temb_act, x2=None):
        """x2: optional second PF tensor -- the block's input is the channel concatenation [x | x2] (a skip connection), which is
        never materialised: GroupNorm and the shortcut read both tensors in place."""
        from .pf import PF, conv_pf, conv3x3_res_pf, conv3x3_res_fusable, groupnorm_pf2
        cout = self.conv2.out_channels
        fuse = self.conv_shortcut is not None and conv3x3_res_fusable(x, cout) and (x2 is None or x2.C % 64 == 0)
        if x2 is not None and not fuse:
            x, x2 = PF(torch.cat([x.buf, x2.buf], dim=1), x.B, x.H, x.W, x.C + x2.C), None
        n1 = groupnorm_pf2(x, x2, self.norm1.weight, self.norm1.bias, self.norm1.num_groups, self.norm1.eps, act=True)
        rowbias = temb_act.rows[id(self)] if isinstance(temb_act, TembRows) else _lin(temb_act, self.time_emb_proj).contiguous()
        h = conv_pf(n1, _pw(self.conv1), self.conv1.bias, rowbias=rowbias, gn_only=True)       # read by norm2 and nothing else: no border zeroing when it wrote records
        h = _gn_pf(h, self.norm2)
        if self.conv_shortcut is None:
            return conv_pf(h, _pw(self.conv2), self.conv2.bias, resid=x)                 # residual add in the GEMM epilogue
        if not fuse:
            sc = conv_pf(x, _pw(self.conv_shortcut), self.conv_shortcut.bias, ksize=1)
            return conv_pf(h, _pw(self.conv2), self.conv2.bias, resid=sc)
        from .pf import cached
        c = cached(self, "_gsw_res", (self.conv2.weight, self.conv2.bias, self.conv_shortcut.weight, self.conv_shortcut.bias),
                   lambda: (torch.cat([_pw(self.conv2), self.conv_shortcut.weight.detach()[:, :, 0, 0]], dim=1).contiguous(),      # [N, 9*C | C_shortcut]
                            (self.conv2.bias.detach() + self.conv_shortcut.bias.detach()).contiguous()))                           # summed biases
        return conv3x3_res_pf(h, c[0], c[1], x1=x, x2=x2)                                # conv2 + conv_shortcut in one GEMM


def _padded_ctx(ctx: torch.Tensor):
    """Context tokens zero-padded to a multiple of 64 keys for the attention kernel (77 -> 128), computed once per context tensor:
    every cross-attention layer of every step receives the same object."""
    n = ctx.shape[1]
    if n % 64 == 0:
        return ctx, n
    cached = getattr(ctx, "_gsw_pad", None)              # (tensor version at padding time, padded copy)
    if cached is not None and cached[1].device == ctx.device and cached[1].dtype == ctx.dtype:
        if cached[0] == ctx._version:
            return cached[1], n
        # the context was edited in place: refresh the padded copy IN PLACE too, so that its address -- which a captured HIP graph of the
        # forward (graph.py) has baked in -- stays valid; in-place edits of ctx bump _version, the copy_ below bumps the pad's own
        cached[1][:, :n].copy_(ctx)
        ctx._gsw_pad = (ctx._version, cached[1])
        return cached[1], n
    pad = F.pad(ctx, (0, 0, 0, (-n) % 64)).contiguous()
    ctx._gsw_pad = (ctx._version, pad)
    return pad, n


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads, head_dim):
        super().__init__()
        if head_dim is None:                     # SD 1.x: a fixed NUMBER of heads (8), head_dim = dim / heads (40 / 80 / 160)
            head_dim = dim // heads
        inner = heads * head_dim
        self.heads = heads
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(ctx_dim, inner, bias=False)
        self.to_v = nn.Linear(ctx_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim)])

    def context_kv(self, src: torch.Tensor):
        """Cross-attention keys / values^T of a (padded) context tensor: they depend on the context only, so they are computed once per
        (context tensor, layer) and reused by every step of a sampling / inversion loop.  The cache lives ON the context tensor (it dies with
        it) and is keyed by the tensor's and the weights' version counters, so in-place edits recompute it -- INTO the existing buffers: a
        captured HIP graph of the forward (graph.py) reads K / V^T at fixed addresses."""
        store = getattr(src, "_gsw_kv", None)
        if store is None:
            store = {}
            src._gsw_kv = store
        ver = (src._version, self.to_k.weight._version, self.to_v.weight._version, self.to_k.weight.data_ptr(), self.to_v.weight.data_ptr())
        ent = store.get(id(self))
        if ent is None or ent[0] != ver:
            ent = (ver, _lin(src, self.to_k, out=None if ent is None else ent[1]), _lin_t(src, self.to_v, out=None if ent is None else ent[2]))
            store[id(self)] = ent
        return ent[1], ent[2]

    def forward_ln(self, x, stat, norm: nn.LayerNorm, ctx=None, resid=None, project: bool = True):
        """attn(LayerNorm(x)) + resid with the LayerNorm folded into the projections that consume it (pf.gemm_ln): x is the RAW residual stream,
        stat its per-row (rstd, -rstd mean).  Self-attention: q | k and V^T; cross-attention: q (keys / values come from the context).
        project=False: the attention output BEFORE to_out (the caller hands it, to_out and the residual to the launch that consumes them: fused_sublayer)."""
        from .pf import attention, cached, fold_ln_weights, gemm_ln
        b, n, _ = x.shape
        inner = self.to_q.out_features
        if ctx is None:
            fq = cached(self, "_gsw_ln_qk", (self.to_q.weight, self.to_k.weight, norm.weight, norm.bias),
                        lambda: fold_ln_weights(torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach()], dim=0), None, norm.weight, norm.bias))
            fv = cached(self, "_gsw_ln_v", (self.to_v.weight, norm.weight, norm.bias), lambda: fold_ln_weights(self.to_v.weight.detach(), None, norm.weight, norm.bias))
            qk = gemm_ln(x, stat, *fq)
            vt = gemm_ln(x, stat, *fv, mode="trans", tokens=n)
            o = attention(qk[..., :inner], qk[..., inner:], vt, self.heads)
        else:
            src, valid = _padded_ctx(ctx)
            fq = cached(self, "_gsw_ln_q", (self.to_q.weight, norm.weight, norm.bias), lambda: fold_ln_weights(self.to_q.weight.detach(), None, norm.weight, norm.bias))
            k_ctx, vt_ctx = self.context_kv(src)
            o = attention(gemm_ln(x, stat, *fq), k_ctx, vt_ctx, self.heads, valid_keys=valid)
        if not project:
            return o
        return _lin(o, self.to_out[0], resid, rowstats=True)

    def fused_sublayer(self, x, stat, norm: nn.LayerNorm, ctx, eps_next: Optional[float] = None, pre=None):
        """x + attn(LayerNorm(x), ctx) as ONE launch (xattn.py / csrc/gswm_xattn.hip): x [B, S, 320] raw residual stream, stat its (rstd, -rstd mean), ctx
        [B or 2B, 77, D] (2B: classifier-free guidance on shared latents -> [2B, S, 320]).  eps_next: leave the statistics of the new rows for the next LayerNorm.
        pre = (o, to_out): the same launch also runs the output projection of the self-attention in front -- x is then that projection's RESIDUAL, o the
        attention output, and the stream x + to_out(o) the sublayer works on exists in registers only (stat is not needed)."""
        from . import xattn
        blob, v, idx = xattn.context_operands(self, norm, ctx, x.dtype)
        if pre is not None:
            return xattn.fused(x, None, blob, v, idx, ctx.shape[0], self.heads, eps_out=eps_next, pre_o=pre[0], pre_w=xattn.out_projection_operand(pre[1], x.dtype),
                               pre_eps=norm.eps)
        return xattn.fused(x, stat, blob, v, idx, ctx.shape[0], self.heads, eps_out=eps_next)

    def cross_dup(self, x, ctx, *, stat=None, norm: Optional[nn.LayerNorm] = None):
        """Classifier-free guidance with shared latents: x [B, S, C] holds the queries' input ONCE, ctx [2B, 77, D] = (uncond | text) contexts.  The
        query projection runs on B rows; the 77-key attention once per context half, both writing one [2B, S, C] tensor; -> attention output (before
        the output projection).  stat / norm: LayerNorm folded into the projection (x is the raw residual stream), else x is already normalised."""
        from .pf import attention, cached, fold_ln_weights, gemm_ln
        B = x.shape[0]
        src, valid = _padded_ctx(ctx)
        k_ctx, vt_ctx = self.context_kv(src)
        if stat is not None:
            fq = cached(self, "_gsw_ln_q", (self.to_q.weight, norm.weight, norm.bias), lambda: fold_ln_weights(self.to_q.weight.detach(), None, norm.weight, norm.bias))
            q = gemm_ln(x, stat, *fq)
        else:
            q = _lin(x, self.to_q)
        o = torch.empty((2 * B, x.shape[1], self.to_q.out_features), dtype=x.dtype, device=x.device)
        attention(q, k_ctx[:B], vt_ctx[:B], self.heads, valid_keys=valid, out=o[:B])
        attention(q, k_ctx[B:], vt_ctx[B:], self.heads, valid_keys=valid, out=o[B:])
        return o

    def ln_foldable(self, x, ctx=None) -> bool:
        from .pf import attention_ok
        n = x.shape[1]
        sk = n if ctx is None else (ctx.shape[1] + 63) // 64 * 64
        return (OWN_ATTENTION and FUSED_KERNELS and n % 8 == 0 and self.to_q.bias is None and self.to_k.bias is None and self.to_v.bias is None
                and attention_ok(x, self.heads, self.to_q.out_features // self.heads, n, sk) and (ctx is None or CACHE_CONTEXT_KV)
                and _own_gemm_ok(x, self.to_q.in_features, self.to_q.out_features))

    def forward(self, x, ctx=None, resid=None):
        """resid: added to the output projection (in its GEMM epilogue on the own path): `x + attn(norm(x))` of the transformer block"""
        b, n, _ = x.shape
        if OWN_ATTENTION and FUSED_KERNELS:
            from .pf import attention, attention_ok
            src, valid = (x, n) if ctx is None else _padded_ctx(ctx)
            if attention_ok(x, self.heads, self.to_q.out_features // self.heads, n, src.shape[1]):
                # hand-written flash-attention kernel (self- and cross-attention); the value projection is computed transposed
                # (V^T = W_v src^T, one GEMM either way) because the kernel consumes V^T tiles.  Padded context rows are zero and
                # masked by `valid`.
                if ctx is not None and CACHE_CONTEXT_KV:
                    k_ctx, vt_ctx = self.context_kv(src)
                    o = attention(_lin(x, self.to_q), k_ctx, vt_ctx, self.heads, valid_keys=valid)
                    return _lin(o, self.to_out[0], resid)
                inner = self.to_q.out_features
                if (ctx is None and FUSED_QK and FUSED_QKV and self.to_q.bias is None and self.to_k.bias is None and self.to_v.bias is None
                        and (2 * inner) % 160 == 0 and n % 8 == 0 and _own_gemm_ok(x, self.to_q.in_features, 3 * inner)):
                    from .pf import cached, gemm_qkv
                    wqkv = cached(self, "_gsw_wqkv", (self.to_q.weight, self.to_k.weight, self.to_v.weight),
                                  lambda: torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach(), self.to_v.weight.detach()], dim=0).contiguous())
                    if wqkv.dtype == x.dtype:
                        qk, vt = gemm_qkv(x.contiguous(), wqkv, 2 * inner)
                        o = attention(qk[..., :inner], qk[..., inner:], vt, self.heads, valid_keys=valid)
                        return _lin(o, self.to_out[0], resid)
                vt = _lin_t(src, self.to_v)
                if ctx is None and FUSED_QK:
                    # self-attention: q and k from ONE GEMM over x (x is read once); the kernel takes them as column slices
                    from .pf import cached
                    wqk = cached(self, "_gsw_wqk", (self.to_q.weight, self.to_k.weight), lambda: torch.cat([self.to_q.weight.detach(), self.to_k.weight.detach()], dim=0).contiguous())
                    if _own_gemm_ok(x, wqk.shape[1], wqk.shape[0]):
                        from .pf import gemm
                        qk = gemm(x.contiguous(), wqk, None)
                    else:
                        qk = F.linear(x, wqk)
                    inner = self.to_q.out_features
                    o = attention(qk[..., :inner], qk[..., inner:], vt, self.heads, valid_keys=valid)
                else:
                    o = attention(_lin(x, self.to_q), _lin(src, self.to_k), vt, self.heads, valid_keys=valid)
                return _lin(o, self.to_out[0], resid)
            if x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
                _note_fallback(f"attention head_dim={self.to_q.out_features // self.heads} Sq={n} Sk={src.shape[1]}: torch SDPA")
        ctx = x if ctx is None else ctx
        q = _lin(x, self.to_q).view(b, n, self.heads, -1).transpose(1, 2)
        k = _lin(ctx, self.to_k).view(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        v = _lin(ctx, self.to_v).view(b, ctx.shape[1], self.heads, -1).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        return _lin(o.transpose(1, 2).reshape(b, n, -1), self.to_out[0], resid)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        inner = self.proj.out_features // 2
        if _own_gemm_ok(x, self.proj.in_features, self.proj.out_features) and inner % 80 == 0:
            from .pf import gemm, pack_geglu_weight, cached     # value * gelu(gate) in the GEMM epilogue: no [M, 2I] intermediate
            _wb(self.proj, x)
            c = cached(self, "_gsw_geglu", (self.proj.weight, self.proj.bias), lambda: pack_geglu_weight(self.proj.weight.detach(), self.proj.bias.detach()))
            return gemm(x.contiguous(), c[0], c[1], mode="geglu")
        y = self.proj(x)
        if FUSED_KERNELS and y.is_cuda and y.is_contiguous() and (y.shape[-1] // 2) % 8 == 0:
            from . import codec
            return codec.geglu(y)
        h, gate = y.chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), nn.Linear(dim * mult, dim)])

    def forward(self, x, resid=None, rowstats: bool = False):
        return _lin(self.net[0](x), self.net[2], resid, rowstats=rowstats)

    def forward_ln(self, x, stat, norm: nn.LayerNorm, resid=None):
        """ff(LayerNorm(x)) + resid with the LayerNorm folded into the GEGLU projection (pf.gemm_ln)"""
        from .pf import cached, fold_ln_weights, gemm_ln
        proj = self.net[0].proj
        f = cached(self, "_gsw_ln_ff1", (proj.weight, proj.bias, norm.weight, norm.bias), lambda: fold_ln_weights(proj.weight.detach(), proj.bias, norm.weight, norm.bias, geglu=True))
        return _lin(gemm_ln(x, stat, *f, mode="geglu"), self.net[2], resid, rowstats=True)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads, head_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, dim, heads, head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, ctx_dim, heads, head_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def dup_ok(self, x, ctx) -> bool:
        """forward(x, ctx, dup=True) is available: the fused path with the own attention kernel and cached context K / V"""
        from .pf import attention_ok
        a = self.attn2
        return (FUSED_KERNELS and OWN_ATTENTION and CACHE_CONTEXT_KV and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.is_contiguous()
                and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1536 and ctx.shape[0] == 2 * x.shape[0]
                and attention_ok(x, a.heads, a.to_q.out_features // a.heads, x.shape[1], (ctx.shape[1] + 63) // 64 * 64))

    def _cross_sublayer(self, x, ctx, dup: bool, one_launch: bool):
        """x + attn2(norm2(x), ctx) on the hand-written path; dup: x holds B rows, ctx 2B -> 2B rows"""
        from .codec import add_layernorm
        from .pf import ln_stat
        st = ln_stat(x, self.norm2.eps) if (one_launch or self.attn2.ln_foldable(x, ctx[: x.shape[0]] if dup else ctx)) else None
        if st is not None and one_launch:
            return self.attn2.fused_sublayer(x, st, self.norm2, ctx, eps_next=self.norm3.eps)
        if dup:
            if st is not None:
                o = self.attn2.cross_dup(x, ctx, stat=st, norm=self.norm2)
            else:
                _, n = add_layernorm(x, None, self.norm2.weight, self.norm2.bias, self.norm2.eps)
                o = self.attn2.cross_dup(n, ctx)
            return _lin(o, self.attn2.to_out[0], torch.cat([x, x], dim=0), rowstats=True)
        if st is not None:
            return self.attn2.forward_ln(x, st, self.norm2, ctx, resid=x)
        _, n = add_layernorm(x, None, self.norm2.weight, self.norm2.bias, self.norm2.eps)
        return self.attn2(n, ctx, resid=x)

    def forward(self, x, ctx, dup: bool = False):
        """dup (classifier-free guidance, see UNet2DCondition.forward): x [B, S, C], ctx [2B, ...] -> [2B, S, C]; self-attention and the cross-attention
        queries are computed once on the B rows the two halves share."""
        if dup and not self.dup_ok(x, ctx):
            x, dup = torch.cat([x, x], dim=0), False
        if FUSED_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.is_contiguous() and x.shape[-1] % 8 == 0 \
                and x.shape[-1] <= 1536:
            from .codec import add_layernorm
            from .pf import ln_stat
            # the residual adds ride in the output projections' GEMM epilogues.  Each LayerNorm is FOLDED into the projections that consume it when the
            # launch that produced x left row records on it (large batches: pf.ln_stat) -- the normalised tensor is then never written; otherwise it is
            # one read + one write (gsw_add_layernorm)
            from . import xattn
            one_launch = xattn.usable(x, self.attn2, ctx)      # the 320-channel level: norm2 + query projection + 77-key attention + output projection + residual in one kernel
            st = ln_stat(x, self.norm1.eps) if self.attn1.ln_foldable(x) else None
            if st is not None and one_launch and xattn.PRE_ENABLED and tuple(self.attn1.to_out[0].weight.shape) == (xattn.CHANNELS, xattn.CHANNELS):
                # ... and the self-attention's output projection + bias + residual + norm2's statistics as that kernel's prologue: the stream between the two
                # attention sublayers is never stored (dup: the launch writes 2B rows from B)
                o1 = self.attn1.forward_ln(x, st, self.norm1, project=False)
                x = self.attn2.fused_sublayer(x, None, self.norm2, ctx, eps_next=self.norm3.eps, pre=(o1, self.attn1.to_out[0]))
            else:
                if st is not None:
                    x = self.attn1.forward_ln(x, st, self.norm1, resid=x)
                else:
                    _, n = add_layernorm(x, None, self.norm1.weight, self.norm1.bias, self.norm1.eps)
                    x = self.attn1(n, resid=x)
                x = self._cross_sublayer(x, ctx, dup, one_launch)
            inner4 = self.ff.net[2].in_features
            st = ln_stat(x, self.norm3.eps) if (inner4 % 80 == 0 and _own_gemm_ok(x, x.shape[-1], 2 * inner4)) else None
            if st is not None:
                return self.ff.forward_ln(x, st, self.norm3, resid=x)
            _, n = add_layernorm(x, None, self.norm3.weight, self.norm3.bias, self.norm3.eps)
            return self.ff(n, resid=x)
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, ch, ctx_dim, heads, head_dim, layers=1, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.proj_in = nn.Linear(ch, ch)        # use_linear_projection=True (SD 2.x)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(ch, ctx_dim, heads, head_dim) for _ in range(layers)])
        self.proj_out = nn.Linear(ch, ch)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        y = gn_act(x, self.norm, act=False).permute(0, 2, 3, 1).reshape(b, h * w, c)
        y = self.proj_in(y)
        for blk in self.transformer_blocks:
            y = blk(y, ctx)
        y = self.proj_out(y).reshape(b, h, w, c).permute(0, 3, 1, 2)
        return x + y

    def forward_pf(self, x, ctx, dup: bool = False):
        """dup: x holds B images, ctx 2B contexts -> a PF tensor of 2B images (see UNet2DCondition.forward)"""
        from . import xattn
        if xattn.gn_proj_usable(x, self.norm, self.proj_in):
            # the 320-channel level at large batch: GroupNorm + proj_in in one launch, the rows normalised on their way into the matrix pipe (never stored)
            y = xattn.gn_proj(x, self.norm, self.proj_in, eps_next=self.transformer_blocks[0].norm1.eps)
        else:
            y = _lin(_gn_pf(x, self.norm, act=False, tokens=True), self.proj_in, rowstats=True)      # GroupNorm writes dense tokens directly
        for i, blk in enumerate(self.transformer_blocks):
            y = blk(y, ctx, dup=dup and i == 0)
        if dup:
            from .pf import dup_pf
            x = dup_pf(x)
        if _own_gemm_ok(y, self.proj_out.in_features, self.proj_out.out_features):
            from .pf import gemm        # proj_out + residual written straight into the PF tensor's interior rows (x has no other reader)
            w, b = _wb(self.proj_out, y)
            gemm(y.contiguous(), w, b, resid=x.rows, mode="tok2pf", tokens=x.H * x.W, width=x.W, out=x.rows, stats_for=x)
        else:
            x.interior.add_(_lin(y, self.proj_out).view(x.B, x.H, x.W, x.C))
            x.stats = None
        return x


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)

    def forward_pf(self, x):
        from .pf import conv_pf
        return conv_pf(x, _pw(self.conv), self.conv.bias, stride=2)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))

    def forward_pf(self, x):
        from .pf import PF, conv_pf, conv_up2x_pf, conv_up2x_fusable, pack_upsample_weight
        if UPSAMPLE_SUBPIXEL and conv_up2x_fusable(x, self.conv.out_channels):
            from .pf import cached
            w4 = cached(self, "_gsw_up4", (self.conv.weight,), lambda: pack_upsample_weight(self.conv.weight))
            return conv_up2x_pf(x, w4, self.conv.bias)            # 2.25x fewer FLOPs, no upsampled intermediate
        up = PF.zeros(x.B, 2 * x.H, 2 * x.W, x.C, x.buf.dtype, x.buf.device)
        xi, g = x.interior, up.grid
        for dy in (0, 1):
            for dx in (0, 1):
                g[:, 1 + dy:1 + dy + 2 * x.H:2, 1 + dx:1 + dx + 2 * x.W:2, :].copy_(xi)
        return conv_pf(up, _pw(self.conv), self.conv.bias)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, ctx_dim, heads, head_dim, layers, attn, down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(cout, ctx_dim, heads, head_dim) for _ in range(layers)]) if attn else None
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if down else None

    def forward(self, x, temb, ctx, skips):
        for i, r in enumerate(self.resnets):
            x = r(x, temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
            skips.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            skips.append(x)
        return x

    def forward_pf(self, x, temb, ctx, skips, dup_temb=None):
        """dup_temb (first down block under classifier-free guidance with shared latents): x holds B images, ctx 2B contexts; the first resnet and the
        first transformer up to its cross-attention run on B images, everything after on 2B with the doubled time-embedding rows `dup_temb`"""
        from .pf import PF
        for i, r in enumerate(self.resnets):
            x = r.forward_pf(x, temb)
            if self.attentions is not None:
                x = self.attentions[i].forward_pf(x, ctx, dup=dup_temb is not None and i == 0)
                if dup_temb is not None and i == 0:
                    temb = dup_temb
            skips.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0].forward_pf(x)
            skips.append(x)
        return x


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, ctx_dim, heads, head_dim, layers, attn, up):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb))
        self.resnets = nn.ModuleList(rs)
        self.attentions = nn.ModuleList([Transformer2DModel(cout, ctx_dim, heads, head_dim) for _ in range(layers)]) if attn else None
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if up else None

    def forward(self, x, temb, ctx, skips):
        for i, r in enumerate(self.resnets):
            x = r(torch.cat([x, skips.pop()], dim=1), temb)
            if self.attentions is not None:
                x = self.attentions[i](x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x

    def forward_pf(self, x, temb, ctx, skips):
        from .pf import PF
        for i, r in enumerate(self.resnets):
            x = r.forward_pf(x, temb, x2=skips.pop())
            if self.attentions is not None:
                x = self.attentions[i].forward_pf(x, ctx)
        if self.upsamplers is not None:
            x = self.upsamplers[0].forward_pf(x)
        return x


class MidBlock(nn.Module):
    def __init__(self, ch, temb, ctx_dim, heads, head_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb), ResnetBlock2D(ch, ch, temb)])
        self.attentions = nn.ModuleList([Transformer2DModel(ch, ctx_dim, heads, head_dim)])

    def forward(self, x, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](x, temb), ctx), temb)

    def forward_pf(self, x, temb, ctx):
        return self.resnets[1].forward_pf(self.attentions[0].forward_pf(self.resnets[0].forward_pf(x, temb), ctx), temb)


class UNet2DCondition(nn.Module):
    """SD 2.1-base shape by default (865.9 M parameters); `UNet2DCondition.sd15()` gives the SD 1.5 shape of BASELINE config 5."""

    @classmethod
    def sd15(cls) -> "UNet2DCondition":
        """runwayml/stable-diffusion-v1-5 `unet/config.json`: 8 heads everywhere (head_dim 40/80/160/160), cross_attention_dim 768.
        Its proj_in / proj_out are 1x1 convolutions, numerically the linear layers used here (load_diffusers_state_dict squeezes
        the [C, C, 1, 1] weights)."""
        return cls(cross_attention_dim=768, num_heads=(8, 8, 8, 8), head_dim=None)

    def __init__(self, in_channels=4, out_channels=4, block_out_channels: Sequence[int] = (320, 640, 1280, 1280), layers_per_block=2,
                 cross_attention_dim=1024, num_heads: Sequence[int] = (5, 10, 20, 20), head_dim=64,
                 attn_blocks: Sequence[bool] = (True, True, True, False)):
        super().__init__()
        c0 = block_out_channels[0]
        temb = c0 * 4
        self.c0 = c0
        self.conv_in = nn.Conv2d(in_channels, c0, 3, padding=1)
        self.time_embedding = TimestepEmbedding(c0, temb)
        self.down_blocks = nn.ModuleList()
        ch = c0
        n = len(block_out_channels)
        for i, co in enumerate(block_out_channels):
            self.down_blocks.append(DownBlock(ch, co, temb, cross_attention_dim, num_heads[i], co // num_heads[i] if head_dim is None else head_dim,
                                              layers_per_block, attn_blocks[i], i < n - 1))
            ch = co
        self.mid_block = MidBlock(ch, temb, cross_attention_dim, num_heads[-1], head_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        rheads = list(reversed(num_heads))
        rattn = list(reversed(attn_blocks))
        cprev = ch
        for i, co in enumerate(rev):
            cin = rev[min(i + 1, n - 1)]
            self.up_blocks.append(UpBlock(cin, co, cprev, temb, cross_attention_dim, rheads[i], head_dim, layers_per_block + 1, rattn[i], i < n - 1))
            cprev = co
        self.conv_norm_out = nn.GroupNorm(32, c0, eps=1e-5)
        self.conv_out = nn.Conv2d(c0, out_channels, 3, padding=1)

    supports_cfg_dup = True

    def forward(self, x: torch.Tensor, t: torch.Tensor, ctx: torch.Tensor, cfg_dup: bool = False) -> torch.Tensor:
        """x [B,4,h,w], t [B] (or scalar tensor) timesteps, ctx [B,77,1024] -> model output [B,4,h,w].
        cfg_dup (classifier-free guidance, the reference's `torch.cat([latents] * 2)` of modified_stable_diffusion_gs.pyc): ctx holds 2B contexts
        (uncond | text) for the SAME B latents -> output [2B,4,h,w].  Until the first cross-attention the two halves of the batch are the same
        numbers -- conv_in, the first resnet, the first transformer's GroupNorm / proj_in / self-attention / query projection -- so they are computed
        once on B rows (CFG_SHARED_PREFIX; per-row results are what the doubled batch would compute)."""
        if cfg_dup:
            if ctx.shape[0] != 2 * x.shape[0] or t.numel() not in (1, x.shape[0]):
                raise ValueError("cfg_dup: ctx must hold 2B contexts for the B latents, t one timestep or B")
            if not (CFG_SHARED_PREFIX and self._pf_ok(x) and self._cfg_dup_ok(x, ctx)):
                return self.forward(torch.cat([x, x], dim=0), t if t.numel() == 1 else torch.cat([t, t]), ctx)
        if self._pf_ok(x):
            rows = _temb_rows_from_table(self, t, x) if TEMB_TABLE else None
            if rows is not None:
                return self._forward_pf(x, rows, ctx, cfg_dup)
        if t.dim() == 0:
            t = t.expand(x.shape[0])
        temb = self.time_embedding(timestep_embedding(t, self.c0).to(x.dtype))
        temb = F.silu(temb)                      # every resnet applies SiLU to the same embedding: do it once
        if self._pf_ok(x):
            return self._forward_pf(x, _temb_rows(self, temb), ctx, cfg_dup)
        if USE_PF and FUSED_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16):
            _note_fallback(f"UNet forward on {tuple(x.shape)} {x.dtype}: off the padded-flat path (fp16 / bf16, conv channels % 64, lattice % {1 << (len(self.down_blocks) - 1)}): plain torch modules")
        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            h = blk(h, temb, ctx, skips)
        h = self.mid_block(h, temb, ctx)
        for blk in self.up_blocks:
            h = blk(h, temb, ctx, skips)
        # the scheduler-step / vote kernels index the lattice in C order: hand back plain NCHW even when the convolutions
        # run channels-last
        return self.conv_out(gn_act(h, self.conv_norm_out)).contiguous(memory_format=torch.contiguous_format)


def _unet_pf_ok(self, x: torch.Tensor) -> bool:
    if not (USE_PF and FUSED_KERNELS and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16)):
        return False
    ok = getattr(self, "_pf_shapes_ok", None)
    if ok is None:
        ok = all(m.in_channels % 64 == 0 and m.out_channels % 64 == 0 for n, m in self.named_modules()
                 if isinstance(m, nn.Conv2d) and n not in ("conv_in", "conv_out"))
        self._pf_shapes_ok = ok
    n_down = len(self.down_blocks) - 1
    return ok and x.shape[-1] % (1 << n_down) == 0 and x.shape[-2] % (1 << n_down) == 0


def _edge_conv_weights(self):
    """conv_in (4 -> 320) and conv_out (320 -> 4) for the PF GEMM: the 4-channel side is zero-padded to one 64-wide tile."""
    from .pf import cached, pack_conv_weight
    w_in, w_out = self.conv_in.weight, self.conv_out.weight

    def build():
        wi = torch.zeros((w_in.shape[0], 64, 3, 3), dtype=w_in.dtype, device=w_in.device)
        wi[:, : w_in.shape[1]] = w_in.detach()
        wo = torch.zeros((64, w_out.shape[1], 3, 3), dtype=w_out.dtype, device=w_out.device)
        wo[: w_out.shape[0]] = w_out.detach()
        bo = torch.zeros(64, dtype=w_out.dtype, device=w_out.device)
        bo[: w_out.shape[0]] = self.conv_out.bias.detach()
        return pack_conv_weight(wi), pack_conv_weight(wo), bo

    return cached(self, "_gsw_edge", (w_in, w_out, self.conv_out.bias), build)


def _temb_rows(self, temb: torch.Tensor):
    """Every resnet's time_emb_proj(temb) from one GEMM over the row-concatenated weights -> TembRows (or temb itself off the engine)."""
    resnets = getattr(self, "_gsw_resnets", None)
    if resnets is None:
        resnets = self._gsw_resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
    n_tot = sum(r.time_emb_proj.out_features for r in resnets)
    if not resnets or not _own_gemm_ok(temb, temb.shape[-1], n_tot) or any(r.time_emb_proj.out_features % 8 for r in resnets):
        return temb
    from .pf import cached, gemm
    params = tuple(p for r in resnets for p in (r.time_emb_proj.weight, r.time_emb_proj.bias))
    wcat, bcat = cached(self, "_gsw_temb_cat", params, lambda: (torch.cat([r.time_emb_proj.weight.detach() for r in resnets], dim=0).contiguous(),
                                                                  torch.cat([r.time_emb_proj.bias.detach() for r in resnets], dim=0).contiguous()))
    if wcat.dtype != temb.dtype:
        return temb
    rb = gemm(temb.contiguous(), wcat, bcat)
    rows, off = {}, 0
    for r in resnets:
        n = r.time_emb_proj.out_features
        rows[id(r)] = rb[:, off:off + n]
        off += n
    return TembRows(temb, rows, rb)


TEMB_TABLE = True     # integer timesteps: the whole time-embedding chain (sinusoid -> linear -> SiLU -> linear -> SiLU -> every resnet's time_emb_proj) is a
                      # function of t alone -- tabulated once over the training timesteps, a forward gathers its rows (one launch instead of ~20, and the
                      # 50 MB of projection weights are not streamed per forward)
NUM_TRAIN_TIMESTEPS = 1000   # default length of the table; a model built from a checkpoint carries its scheduler's value as `model.num_train_timesteps`
                             # (extract.Models sets it from scheduler/scheduler_config.json), and ddim.DDIMSchedule refuses timesteps outside the range on the host


def _temb_rows_from_table(self, t: torch.Tensor, x: torch.Tensor):
    """TembRows gathered from the per-model table [num_train_timesteps, sum of the resnets' channels], or None when t is not an integer tensor / the
    projections do not run on the engine.  The table is as long as the model's `num_train_timesteps` (the scheduler config's; 1000 by default).  The
    schedules of ddim.py validate their timesteps against that range on the host; a device tensor outside it would be clamped by gsw_gather_rows."""
    if t.dtype not in (torch.int64, torch.int32) or not t.is_cuda or t.numel() not in (1, x.shape[0]):
        return None
    resnets = getattr(self, "_gsw_resnets", None)
    if resnets is None:
        resnets = self._gsw_resnets = [m for m in self.modules() if isinstance(m, ResnetBlock2D)]
    if not resnets or any(r.time_emb_proj.out_features % 8 for r in resnets):
        return None
    te = self.time_embedding
    if te.linear_1.weight.dtype != x.dtype or te.linear_1.weight.device != x.device:
        return None
    from .pf import cached
    from . import _native as N
    from .codec import _stream_ptr
    params = (te.linear_1.weight, te.linear_1.bias, te.linear_2.weight, te.linear_2.bias) + tuple(p for r in resnets for p in (r.time_emb_proj.weight, r.time_emb_proj.bias))

    def build():
        with torch.no_grad():
            tt = torch.arange(int(getattr(self, "num_train_timesteps", NUM_TRAIN_TIMESTEPS)), device=x.device)
            temb = F.silu(te(timestep_embedding(tt, self.c0).to(x.dtype)))
            rows = _temb_rows(self, temb)
            if not isinstance(rows, TembRows):
                return None
            return rows.table.contiguous()

    table = cached(self, "_gsw_temb_table", params, build)
    if table is not None and table.shape[0] != int(getattr(self, "num_train_timesteps", NUM_TRAIN_TIMESTEPS)):      # the scheduler length changed after the table was built
        self._gsw_temb_table = None
        table = cached(self, "_gsw_temb_table", params, build)
        from . import graph
        graph.weights_changed()      # a captured forward gathers from the OLD table's address: every graph entry must re-capture
    if table is None:
        return None
    B, n_tot = x.shape[0], table.shape[1]
    idx = t if t.dtype == torch.int64 else t.to(torch.int64)
    out = torch.empty((B, n_tot), dtype=table.dtype, device=table.device)
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_gather_rows(table.data_ptr(), n_tot * table.element_size(), table.shape[0], idx.data_ptr(), 0 if idx.numel() == 1 else 1,
                                        out.data_ptr(), n_tot * table.element_size(), B, n_tot * table.element_size(), _stream_ptr()))
    rows, off = {}, 0
    for r in resnets:
        n = r.time_emb_proj.out_features
        rows[id(r)] = out[:, off:off + n]
        off += n
    return TembRows(None, rows, out)


CONV_OUT_DIRECT_MAX_PIXELS = 65536      # conv_out (320 -> 4) as the one-wave-per-16-pixels kernel writing NCHW directly (gsw_conv3x3_pf_nchw); above that the
                                        # LDS-tiled 64-column kernel moves fewer bytes through L2


CFG_SHARED_PREFIX = __import__("os").environ.get("GSW_CFG_SHARED_PREFIX", "1") != "0"      # classifier-free guidance: the part of a forward that does not see the context runs once for both halves of the batch


def _unet_cfg_dup_ok(self, x: torch.Tensor, ctx: torch.Tensor) -> bool:
    b0 = self.down_blocks[0]
    if b0.attentions is None or not isinstance(b0.attentions[0].transformer_blocks[0], BasicTransformerBlock):
        return False
    tokens = torch.empty((x.shape[0], x.shape[2] * x.shape[3], b0.attentions[0].proj_in.out_features), dtype=x.dtype, device="meta")
    blk = b0.attentions[0].transformer_blocks[0]
    a = blk.attn2
    from .pf import ATTN_HEAD_DIMS
    return (FUSED_KERNELS and OWN_ATTENTION and CACHE_CONTEXT_KV and OWN_GEMM and tokens.shape[-1] % 8 == 0 and tokens.shape[-1] <= 1536
            and (a.to_q.out_features // a.heads) in ATTN_HEAD_DIMS and _own_gemm_ok(x, a.to_q.in_features, a.to_q.out_features))


def _dup_temb(temb):
    """time-embedding rows of B images -> of the 2B images of the doubled batch"""
    if isinstance(temb, TembRows):
        tab = torch.cat([temb.table, temb.table], dim=0)
        rows, off = {}, 0
        for k, v in temb.rows.items():           # (insertion order = column order)
            n = v.shape[1]
            rows[k] = tab[:, off:off + n]
            off += n
        return TembRows(None if temb.act is None else torch.cat([temb.act, temb.act], dim=0), rows, tab)
    return torch.cat([temb, temb], dim=0)


def _unet_forward_pf(self, x: torch.Tensor, temb, ctx: torch.Tensor, cfg_dup: bool = False) -> torch.Tensor:
    from .pf import PF, conv_pf, cached, pack_conv_weight, dup_pf
    from . import _native as N
    from .codec import _dt, _stream_ptr
    w_in, w_out, b_out = _edge_conv_weights(self)
    B, cin, H, W = x.shape
    xin = PF.empty(B, H, W, 64, x.dtype, x.device)
    xc = x.contiguous()
    with torch.cuda.device(x.device):
        N.check(N.lib().gsw_nchw_to_pf(xc.data_ptr(), xin.rows.data_ptr(), B, cin, H, W, 64, _dt(x.dtype), _stream_ptr()))
    h = conv_pf(xin, w_in, self.conv_in.bias)
    skips = [h]
    if cfg_dup:
        temb2 = _dup_temb(temb)
        h = self.down_blocks[0].forward_pf(h, temb, ctx, skips, dup_temb=temb2)
        skips[0] = dup_pf(skips[0])              # the conv_in skip is consumed by the last up-block resnet, on 2B images
        temb, B = temb2, 2 * B
        rest = self.down_blocks[1:]
    else:
        rest = self.down_blocks
    for blk in rest:
        h = blk.forward_pf(h, temb, ctx, skips)
    h = self.mid_block.forward_pf(h, temb, ctx)
    for blk in self.up_blocks:
        h = blk.forward_pf(h, temb, ctx, skips)
    hn = _gn_pf(h, self.conv_norm_out, act=True)
    nout = self.conv_out.out_channels
    if B * H * W <= CONV_OUT_DIRECT_MAX_PIXELS and nout <= 16 and hn.C % 32 == 0 and self.conv_out.bias is not None:
        wo = cached(self.conv_out, "_gsw_packed_direct", (self.conv_out.weight,), lambda: pack_conv_weight(self.conv_out.weight.detach()))
        y = torch.empty((B, nout, H, W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            N.check(N.lib().gsw_conv3x3_pf_nchw(hn.rows.data_ptr(), wo.data_ptr(), self.conv_out.bias.data_ptr(), y.data_ptr(), B, H, W, hn.C, nout,
                                                _dt(x.dtype), _stream_ptr()))
        return y
    y = conv_pf(hn, w_out, b_out)
    return y.interior[..., :nout].permute(0, 3, 1, 2).contiguous()


def _unet_prepare_context(self, ctx: torch.Tensor) -> None:
    """Bring the per-context caches (padded copy, every cross-attention layer's K / V^T) up to date for `ctx` without running a forward --
    in place when they exist (graph.py calls this after overwriting a captured graph's static context buffer)."""
    if not (OWN_ATTENTION and FUSED_KERNELS and CACHE_CONTEXT_KV and ctx.is_cuda):
        return
    from .pf import attention_ok
    src, _ = _padded_ctx(ctx)
    for blk in self.modules():
        if isinstance(blk, BasicTransformerBlock):
            a = blk.attn2
            if attention_ok(ctx, a.heads, a.to_q.out_features // a.heads, 1, src.shape[1]):
                a.context_kv(src)
            from . import xattn
            if xattn.ENABLED and a.to_q.in_features == xattn.CHANNELS and ctx.shape[1] <= xattn.MAX_KEYS and ctx.dtype in (torch.float16, torch.bfloat16):
                xattn.context_operands(a, blk.norm2, ctx, ctx.dtype)      # the one-launch cross-attention's per-context fragment streams
                if xattn.PRE_ENABLED and tuple(blk.attn1.to_out[0].weight.shape) == (xattn.CHANNELS, xattn.CHANNELS):
                    xattn.out_projection_operand(blk.attn1.to_out[0], ctx.dtype)      # ... and the prologue's (the self-attention's output projection; per layer, not per context)


UNet2DCondition._cfg_dup_ok = _unet_cfg_dup_ok
UNet2DCondition.prepare_context = _unet_prepare_context
UNet2DCondition._pf_ok = _unet_pf_ok
UNet2DCondition._forward_pf = _unet_forward_pf


def synthetic_init_(model: nn.Module, seed: int = 0, out_scale: float = 1.0) -> nn.Module:
    """Deterministic synthetic weights (no checkpoint is reachable from here): fan-in scaled normal for matrices/convs,
    ones/zeros for norms.  Residual-branch output layers are down-scaled so activations stay O(1) through ~60 residual
    adds, which keeps fp16 in range like a trained network does."""
    g = torch.Generator().manual_seed(seed)
    n_res = 0
    for name, p in model.named_parameters():
        if p.dim() >= 2:
            fan_in = p[0].numel()
            std = (1.0 / fan_in) ** 0.5
            if name.endswith(("conv2.weight", "to_out.0.weight", "net.2.weight", "proj_out.weight")):
                std *= 0.3
                n_res += 1
            if name == "conv_out.weight":
                std *= out_scale
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * std)
        else:
            with torch.no_grad():
                p.fill_(1.0 if name.endswith("weight") and "norm" in name else 0.0)
    return model


def load_diffusers_state_dict(model: nn.Module, weight_dir: str) -> nn.Module:
    """Load the UNet weights of a diffusers-layout directory (`<weight_dir>/unet/diffusion_pytorch_model.*`: safetensors, sharded safetensors or .bin,
    checkpoint.load_component_state_dict); parameter names match this module 1:1, strictly both ways."""
    from .checkpoint import load_component_state_dict
    sd = dict(load_component_state_dict(weight_dir, "unet"))
    own = dict(model.named_parameters())
    for k, v in list(sd.items()):               # SD 1.x: proj_in / proj_out stored as 1x1 convolutions
        if k in own and v.dim() == 4 and own[k].dim() == 2 and v.shape[2:] == (1, 1):
            sd[k] = v[:, :, 0, 0]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"state dict mismatch: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
    from .graph import weights_changed
    weights_changed()                            # captured graphs of this model (graph.GraphedEpsModel) re-check the parameters before their next replay
    return model


def count_cfg_shared_prefix_flops(model: nn.Module, h: int = 64, w: int = 64) -> int:
    """FLOPs per image of the part of a forward that does not see the context (conv_in, the first resnet, the first transformer's proj_in, self-attention
    and cross-attention query projection): what a cfg_dup forward executes once for two rows of the guidance batch."""
    from torch.utils.flop_counter import FlopCounterMode
    import copy
    m = copy.deepcopy(model).to("meta")
    dt = next(m.parameters()).dtype
    b0 = m.down_blocks[0]
    if b0.attentions is None:
        return 0
    tr = b0.attentions[0]
    blk = tr.transformer_blocks[0]
    x = torch.empty(1, m.conv_in.in_channels, h, w, device="meta", dtype=dt)
    temb = torch.empty(1, m.time_embedding.linear_2.out_features, device="meta", dtype=dt)
    with FlopCounterMode(display=False) as fc:
        hh = m.conv_in(x)
        hh = b0.resnets[0](hh, temb)
        y = tr.proj_in(tr.norm(hh).permute(0, 2, 3, 1).reshape(1, h * w, -1))
        y = y + blk.attn1(blk.norm1(y))
        blk.attn2.to_q(blk.norm2(y))
    return int(fc.get_total_flops())


def count_flops_per_image(model: nn.Module, h: int = 64, w: int = 64, ctx_len: int = 77, ctx_dim: Optional[int] = None) -> int:
    """Forward FLOPs for one image (2 x MACs of conv / linear / attention), via torch's FlopCounterMode on meta tensors."""
    from torch.utils.flop_counter import FlopCounterMode
    import copy
    m = copy.deepcopy(model).to("meta")
    if ctx_dim is None:
        ctx_dim = next(mod for n, mod in m.named_modules() if n.endswith("attn2")).to_k.in_features
    dt = next(m.parameters()).dtype
    x = torch.empty(1, 4, h, w, device="meta", dtype=dt)
    t = torch.empty(1, device="meta")
    c = torch.empty(1, ctx_len, ctx_dim, device="meta", dtype=dt)
    with FlopCounterMode(display=False) as fc:
        m(x, t, c)
    return int(fc.get_total_flops())
