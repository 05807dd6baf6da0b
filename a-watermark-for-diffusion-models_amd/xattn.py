"""The cross-attention sublayer of a transformer block as ONE launch (csrc/gswm_xattn.hip, `gsw_xattn_fused`):

    x' = x + to_out(softmax(to_q(LayerNorm(x)) K^T / sqrt(d)) V) + b_out,        K, V = to_k(ctx), to_v(ctx)

Reference call site: extract.py:66-69 / modified_stable_diffusion_gs.pyc run diffusers' UNet2DConditionModel; this is its
`BasicTransformerBlock.attn2` with `norm2` in front and the residual behind.

Both sides of the softmax are linear in things that do not change from step to step of a sampling / inversion loop -- the weights and the
context -- so per (context, head) the host derives two matrices ONCE (fp32 products of the fp16 weights, one rounding):

    A_h = K_h Wq_h [keys, C]    scores      S_h = LayerNorm(x) A_h^T / sqrt(d)
    B_h = Wo_h V_h^T [C, keys]  output      x'  = x + b_out + sum_h softmax(S_h) B_h^T

LayerNorm is folded: A' = A diag(gamma) * scale * log2(e) with its rows centred over the channels (so x A'^T = (x - mean) A'^T: no rank-one correction for the
mean), v = (A beta) * scale * log2(e), so that S = rstd (x A'^T) + v in the exponent's base 2; padding keys get v = -inf.  `context_operands` stores A' and B in the order the kernel
consumes them: a stream of 1 KiB MFMA fragments (110 per head); the layout algebra is in csrc/gswm_xattn.hip and restated by tests/test_xattn_host.py.

Two more launches of the same transposed-stream family live here, both built on `pack_out_projection` (a 320 x 320 linear layer + bias as 21 chunks of fragments):
  * `fused(..., pre_o=, pre_w=)` (`gsw_xattn_fused_pre`): the self-attention's output projection + residual + norm2's statistics as the cross-attention launch's PROLOGUE
    (diffusers' `BasicTransformerBlock.forward`: `attn1(...) + hidden_states` followed by `attn2(norm2(...)) + hidden_states`) -- the stream between the two is never stored;
  * `gn_proj` (`gsw_gn_proj_tokens`): GroupNorm + `proj_in` at the entry of a transformer (`Transformer2DModel.forward`) from the padded-flat output of the resnet in front and its
    column records -- the normalised tokens are never stored.
"""
from __future__ import annotations

import ctypes as _C
from typing import Optional, Tuple

import torch

from . import _native as N
from .codec import _dt, _stream_ptr

CHANNELS = 320            # the level gsw_xattn_fused serves (SD 2.1: 64 x 64 latents; SD 1.5: the first level)
MAX_KEYS = 79             # context tokens; key slot 79 of the second product carries the output bias
KEY_SLOTS = 96            # three 32-key blocks in the first product (the second one uses slots 0..79)
HEAD_ELEMS = 11 * 5120    # 60 + 50 fragments of 512 elements
V_FLOATS = KEY_SLOTS
LOG2E = 1.4426950408889634
ENABLED = __import__("os").environ.get("GSW_XATTN_FUSED", "1") != "0"      # A/B switch: 0 = the three-launch path (query projection, attention kernel, output projection)


def _column_of(nb: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """output column of accumulator row `m` (0..31) of column block `nb` (0..9): lane (row, hlf) then owns columns 32 nb + 16 j + 8 hlf .. + 7 of its row"""
    return 32 * nb + 16 * (m >> 4) + 8 * ((m >> 2) & 1) + 4 * ((m >> 3) & 1) + (m & 3)


def _key_slot_of(kk: torch.Tensor, hlf: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    """key slot (0..79) that value e of lane half hlf holds in key step kk of the second product: the S^T accumulator layout of v_mfma_f32_32x32x16"""
    return 16 * kk + 8 * (e >> 2) + 4 * hlf + (e & 3)


def fold_operands(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, wo: torch.Tensor, bo: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                  ctx: torch.Tensor, heads: int, dtype: torch.dtype):
    """-> (A' [Bc, H, 96, C] in `dtype`, v [Bc, H, 96] fp32, B [Bc, H, C, 80] in `dtype`) of `ctx` [Bc, keys <= 79, D]: the matrices of the module docstring.
    A' = A diag(gamma) scale log2(e) with every row CENTRED (its mean over the channels subtracted before the rounding): x A'^T is then (x - mean(x)) A'^T, the
    LayerNorm's mean needs no rank-one correction and S = rstd (x A'^T) + v.  Padding key slots: A' rows zero, v = -inf, B columns zero; B's slot 79 = b_out for
    the last head (the kernel gives that slot the probability 1)."""
    Bc, n, _ = ctx.shape
    C = wq.shape[1]
    inner = wq.shape[0]
    d = inner // heads
    if n > MAX_KEYS:
        raise ValueError(f"xattn: {n} context tokens, at most {MAX_KEYS}")
    f = torch.float32
    c32 = ctx.to(f)
    K = (c32 @ wk.detach().to(f).t()).view(Bc, n, heads, d).permute(0, 2, 1, 3)          # [Bc, H, n, d]
    V = (c32 @ wv.detach().to(f).t()).view(Bc, n, heads, d).permute(0, 2, 1, 3)
    A = torch.einsum("bhnd,hdc->bhnc", K, wq.detach().to(f).view(heads, d, C))           # [Bc, H, n, C]
    sc = float(d) ** -0.5 * LOG2E
    Ag = A * (gamma.detach().to(f) * sc)
    Ap = torch.zeros((Bc, heads, KEY_SLOTS, C), dtype=dtype, device=ctx.device)
    Ap[:, :, :n] = (Ag - Ag.mean(dim=-1, keepdim=True)).to(dtype)
    v = torch.full((Bc, heads, KEY_SLOTS), float("-inf"), dtype=f, device=ctx.device)
    v[:, :, :n] = (A @ beta.detach().to(f)) * sc
    Bm = torch.zeros((Bc, heads, C, MAX_KEYS + 1), dtype=dtype, device=ctx.device)
    Bm[..., :n] = torch.einsum("bhnd,chd->bhcn", V, wo.detach().to(f).view(C, heads, d)).to(dtype)
    if bo is not None:
        Bm[:, heads - 1, :, MAX_KEYS] = bo.detach().to(dtype)
    return Ap, v, Bm


def pack_stream(Ap: torch.Tensor, v: torch.Tensor, Bm: torch.Tensor):
    """fold_operands' matrices -> (blob [Bc, H * HEAD_ELEMS] in their dtype, v [Bc, H * V_FLOATS] fp32) in the kernel's consumption order"""
    Bc, H, _, C = Ap.shape
    if C != CHANNELS:
        raise ValueError(f"xattn: C = {C}, the kernel serves {CHANNELS}")
    dev = Ap.device
    # first product: fragment 3 ks + kb (16-channel k-step ks, 32-key block kb), lane (m, hlf), value e = A'[32 kb + m][16 ks + 8 hlf + e]
    g1 = Ap.view(Bc, H, 3, 32, C // 16, 2, 8).permute(0, 1, 4, 2, 5, 3, 6).reshape(Bc, H, -1)      # [ks, kb, hlf, m, e]
    # second product: fragment 10 kk + nb (16-slot key step kk, 32-column block nb), lane (m, hlf), value e = B[column_of(nb, m)][key_slot_of(kk, hlf, e)]
    ar = lambda k: torch.arange(k, device=dev)
    cols = _column_of(ar(C // 32)[:, None], ar(32)[None, :])                                     # [nb, m]
    slots = _key_slot_of(ar(5)[:, None, None], ar(2)[None, :, None], ar(8)[None, None, :])         # [kk, hlf, e]
    g2 = Bm[:, :, cols[None, :, None, :, None], slots[:, None, :, None, :]].reshape(Bc, H, -1)    # [kk, nb, hlf, m, e]
    blob = torch.cat([g1, g2], dim=2).reshape(Bc, H * HEAD_ELEMS).contiguous()
    return blob, v.reshape(Bc, H * V_FLOATS).contiguous()


PRE_CHUNKS = 21           # gsw_xattn_fused_pre's prologue stream: twenty 16-channel k-steps of the self-attention's output projection + its bias
PRE_ENABLED = __import__("os").environ.get("GSW_XATTN_PRE", "1") != "0"      # A/B switch: 0 = the self-attention's output projection stays its own launch


def pack_out_projection(w: torch.Tensor, b: Optional[torch.Tensor], dtype: torch.dtype) -> torch.Tensor:
    """Wo [320, 320] (+ bias) of the SELF-attention in front of the sublayer -> the prologue's fragment stream [PRE_CHUNKS * 5120] in consumption order: chunk ks holds
    the ten column blocks of k-step ks -- fragment nb, lane (m, hlf), value e = Wo[column_of(nb, m)][16 ks + 8 hlf + e] -- and chunk 20 the bias in the k = 0 column of
    ten fragments (lane (m, 0), value 0 = b[column_of(nb, m)]), which the kernel multiplies with a unit vector."""
    C = CHANNELS
    if tuple(w.shape) != (C, C):
        raise ValueError(f"xattn: the output projection must be [{C}, {C}]")
    dev = w.device
    ar = lambda k: torch.arange(k, device=dev)
    cols = _column_of(ar(C // 32)[:, None], ar(32)[None, :])                           # [nb, m]
    wp = w.detach().to(dtype)[cols]                                                    # [nb, m, k]
    g = wp.view(C // 32, 32, C // 16, 2, 8).permute(2, 0, 3, 1, 4).reshape(-1)         # [ks, nb, hlf, m, e]
    bias = torch.zeros(C // 32, 2, 32, 8, dtype=dtype, device=dev)
    if b is not None:
        bias[:, 0, :, 0] = b.detach().to(dtype)[cols]
    return torch.cat([g, bias.reshape(-1)]).contiguous()


def out_projection_operand(lin, dtype: torch.dtype) -> torch.Tensor:
    """pack_out_projection of a Linear, cached on the module (keyed by the parameters' versions; an edit recomputes INTO the same buffer: captured graphs read it)"""
    ent = getattr(lin, "_gsw_xattn_pre", None)
    ver = (str(dtype), lin.weight.data_ptr(), lin.weight._version) + ((lin.bias.data_ptr(), lin.bias._version) if lin.bias is not None else ())
    if ent is None or ent[0] != ver:
        new = pack_out_projection(lin.weight, lin.bias, dtype)
        if ent is not None and ent[1].shape == new.shape and ent[1].dtype == new.dtype and ent[1].device == new.device:
            ent[1].copy_(new)
            new = ent[1]
        ent = (ver, new)
        lin._gsw_xattn_pre = ent
    return ent[1]


def run_index(ctx: torch.Tensor) -> Optional[torch.Tensor]:
    """int32 [Bc]: for every context row the first row of the run of identical rows it belongs to (classifier-free guidance hands over B copies of the
    empty prompt's context: their images then share ONE fragment stream in L2).  Computed on the device, no host synchronisation.  None for one row."""
    Bc = ctx.shape[0]
    if Bc == 1:
        return None
    flat = ctx.reshape(Bc, -1)
    new = torch.ones(Bc, dtype=torch.bool, device=ctx.device)
    new[1:] = (flat[1:] != flat[:-1]).any(dim=1)
    pos = torch.arange(Bc, device=ctx.device, dtype=torch.int32)
    return torch.cummax(torch.where(new, pos, torch.zeros_like(pos)), dim=0).values.contiguous()


def context_operands(attn, norm, ctx: torch.Tensor, dtype: torch.dtype):
    """(blob, v, index) of `ctx` [Bc, keys, D] for cross-attention module `attn` behind LayerNorm `norm`: computed once per (context tensor, layer) and
    reused by every step of a loop.  The cache lives ON the context tensor and is keyed by the tensor's and the parameters' version counters; an in-place
    edit recomputes INTO the existing buffers (a captured HIP graph of the forward reads them at fixed addresses)."""
    src = ctx[:1] if (ctx.dim() == 3 and ctx.shape[0] > 1 and ctx.stride(0) == 0) else ctx           # .expand() of one context: one stream
    store = getattr(ctx, "_gsw_xattn", None)
    if store is None:
        store = {}
        ctx._gsw_xattn = store
    params = (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias, norm.weight, norm.bias)
    ver = (ctx._version, str(dtype)) + tuple((q.data_ptr(), q._version) for q in params if q is not None)
    ent = store.get(id(attn))
    if ent is None or ent[0] != ver:
        with torch.no_grad():
            blob, v = pack_stream(*fold_operands(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias,
                                                  norm.weight, norm.bias, src, attn.heads, dtype))
            idx = run_index(src)
        if ent is not None and ent[1].shape == blob.shape and ent[1].dtype == blob.dtype:
            ent[1].copy_(blob)
            ent[2].copy_(v)
            if idx is not None:
                ent[3].copy_(idx)
            ent = (ver, ent[1], ent[2], ent[3])
        else:
            ent = (ver, blob, v, idx)
        store[id(attn)] = ent
    return ent[1], ent[2], ent[3]


def usable(x: torch.Tensor, attn, ctx: torch.Tensor) -> bool:
    """gsw_xattn_fused serves this call: 320 channels, whole 128-row tiles per image, at most 79 context tokens, no projection biases besides to_out's"""
    return (ENABLED and x.is_cuda and x.dim() == 3 and x.dtype in (torch.float16, torch.bfloat16) and x.is_contiguous() and x.shape[-1] == CHANNELS
            and x.shape[1] % 128 == 0 and ctx.dim() == 3 and ctx.shape[1] <= MAX_KEYS and attn.to_q.bias is None and attn.to_k.bias is None
            and attn.to_v.bias is None and attn.to_q.in_features == CHANNELS and attn.to_out[0].out_features == CHANNELS
            and ctx.shape[0] % x.shape[0] == 0 and x.shape[0] * x.shape[1] < (1 << 31) // max(1, ctx.shape[0] // x.shape[0]))


def fused(x: torch.Tensor, stat: Optional[torch.Tensor], blob: torch.Tensor, v: torch.Tensor, index: Optional[torch.Tensor], out_images: int, heads: int,
          eps_out: Optional[float] = None, out: Optional[torch.Tensor] = None, pre_o: Optional[torch.Tensor] = None, pre_w: Optional[torch.Tensor] = None,
          pre_eps: float = 1e-5) -> torch.Tensor:
    """x [xB, S, 320] raw residual stream, stat [xB * S, 2] = (rstd, -rstd mean) of its rows (pf.ln_stat) -> x' [out_images, S, 320]; output image i
    reads x image i % xB and the context stream index[i] (None: stream 0; one stream: every image's).  eps_out: also leave the (rstd, -rstd mean) of the
    NEW rows on the result (`_gsw_lnstat`, what pf.ln_stat returns for the LayerNorm that follows).
    pre_o / pre_w (gsw_xattn_fused_pre): the launch first MAKES the residual stream from what stands in front of the sublayer in a transformer block -- x becomes the
    residual r of the self-attention's output projection, pre_o [xB, S, 320] that attention's output, pre_w out_projection_operand(to_out) -- i.e. it computes
    x1 = r + pre_o Wo^T + b, LayerNorm statistics of x1 with epsilon pre_eps (stat is not read), then x1 + attn(LayerNorm(x1), ctx); x1 is never stored."""
    if not x.is_cuda:
        raise RuntimeError("xattn.fused: device tensors only; there is no CPU fallback")
    xB, S, C = x.shape
    if C != CHANNELS or S % 128 or out_images % xB or x.dtype not in (torch.float16, torch.bfloat16) or not x.is_contiguous():
        raise ValueError("xattn.fused: x must be a contiguous fp16 / bf16 [xB, S % 128 == 0, 320] tensor, out_images a multiple of xB")
    if blob.dtype != x.dtype or blob.dim() != 2 or blob.shape[1] != heads * HEAD_ELEMS or v.dtype != torch.float32 or v.shape != (blob.shape[0], heads * V_FLOATS):
        raise ValueError("xattn.fused: blob / v are not context_operands' output for this head count and dtype")
    pre = pre_o is not None
    if pre:
        if pre_w is None or pre_o.shape != x.shape or pre_o.dtype != x.dtype or not pre_o.is_contiguous() or not pre_o.is_cuda:
            raise ValueError("xattn.fused: pre_o must match x [xB, S, 320] and come with pre_w")
        if pre_w.dtype != x.dtype or pre_w.numel() != PRE_CHUNKS * 5120 or not pre_w.is_contiguous():
            raise ValueError("xattn.fused: pre_w is not out_projection_operand's output for this dtype")
    elif stat is None or stat.dtype != torch.float32 or stat.numel() != 2 * xB * S or not stat.is_contiguous():
        raise ValueError("xattn.fused: stat must be fp32 [xB * S, 2]")
    nctx = blob.shape[0]
    if index is None and nctx not in (1, out_images):
        raise ValueError("xattn.fused: one context stream, or one per output image, or an index")
    if index is None and nctx == out_images and nctx > 1:
        index = torch.arange(out_images, dtype=torch.int32, device=x.device)
    if index is not None and (index.dtype != torch.int32 or index.numel() != out_images or not index.is_contiguous()):
        raise ValueError("xattn.fused: index must be int32 [out_images]")
    y = torch.empty((out_images, S, C), dtype=x.dtype, device=x.device) if out is None else out
    ostat = torch.empty((out_images * S, 2), dtype=torch.float32, device=x.device) if eps_out is not None else None
    from . import pf
    tm = pf.CONV_TIMER
    with torch.cuda.device(x.device):
        e0 = tm.start() if tm is not None else None
        tail = (blob.data_ptr(), blob.shape[1] * blob.element_size(), v.data_ptr(), v.shape[1], index.data_ptr() if index is not None else None, y.data_ptr(),
                ostat.data_ptr() if ostat is not None else None, float(eps_out) if eps_out is not None else 0.0, xB, out_images, S, C, heads, _dt(x.dtype), _stream_ptr())
        if pre:
            N.check(N.lib().gsw_xattn_fused_pre(x.data_ptr(), pre_o.data_ptr(), pre_w.data_ptr(), float(pre_eps), *tail))
        else:
            N.check(N.lib().gsw_xattn_fused(x.data_ptr(), stat.data_ptr(), *tail))
        if tm is not None:
            tm.stop(e0, ("gsw_xattn_kernel", out_images * S, C, heads * (MAX_KEYS + 1), "xattn+pre" if pre else "xattn") if tm.by_shape else "gsw_xattn_kernel",
                    2.0 * out_images * S * C * (heads * (KEY_SLOTS + MAX_KEYS + 1) + (C + 16 if pre else 0)),
                    nbytes=2.0 * ((2 if pre else 1) * xB + out_images) * S * C + blob.numel() * 2.0 + (pre_w.numel() * 2.0 if pre else 0.0))
    if ostat is not None:
        y._gsw_lnstat = (ostat, float(eps_out))
    return y


# ---- GroupNorm + proj_in of a transformer at the 320-channel level as one launch (gsw_gn_proj_tokens, csrc/gswm_xattn.hip): the normalised tokens are never stored
GNPROJ_ENABLED = __import__("os").environ.get("GSW_GN_PROJ_FUSED", "1") != "0"      # A/B switch: 0 = gsw_gn_pf_apply (tokens) + the engine's 320 x 320 GEMM


def gn_proj_usable(x, norm, lin) -> bool:
    """x: a PF tensor (pf.PF) whose producing launch left GroupNorm column records; norm: the transformer's GroupNorm; lin: its proj_in (nn.Linear)"""
    from . import pf
    C = getattr(x, "C", 0)
    return (GNPROJ_ENABLED and C == CHANNELS and x.buf.is_cuda and x.buf.dtype in (torch.float16, torch.bfloat16) and pf.FUSE_GN_STATS and pf._stats_usable(x)
            and not pf._gn_fused_ok(x.B, x.H, x.W, C, norm.num_groups) and x.W % 32 == 0 and (x.H * x.W) % 128 == 0 and (C // norm.num_groups) % 2 == 0 and C % norm.num_groups == 0
            and x.B * x.H * x.W < (1 << 31) and tuple(lin.weight.shape[:2]) == (CHANNELS, CHANNELS) and lin.weight.numel() == CHANNELS * CHANNELS
            and lin.weight.dtype == x.buf.dtype and norm.weight.dtype == x.buf.dtype)


def gn_proj(x, norm, lin, eps_next: Optional[float] = None) -> torch.Tensor:
    """tokens [B, H W, 320] = GroupNorm(x) Wp^T + b from the PF tensor x and the column records of the launch that produced it; eps_next: leave the (rstd, -rstd mean)
    of the token rows on the result for the LayerNorm that follows (`_gsw_lnstat`, what pf.ln_stat returns)"""
    from . import pf
    if not x.buf.is_cuda:
        raise RuntimeError("xattn.gn_proj: device tensors only; there is no CPU fallback")
    dev, dt, st = x.buf.device, x.buf.dtype, x.stats
    w = out_projection_operand(_as_linear(lin), dt)
    ws = pf._gn_workspace(dev, x.B, norm.num_groups, x.C)
    y = torch.empty((x.B, x.H * x.W, x.C), dtype=dt, device=dev)
    ostat = torch.empty((x.B * x.H * x.W, 2), dtype=torch.float32, device=dev) if eps_next is not None else None
    tm = pf.CONV_TIMER
    with torch.cuda.device(dev):
        N.check(N.lib().gsw_gn_colstats_pairs(st.buf.data_ptr(), st.rows, st.npar, st.blocks, ws.data_ptr(), x.B, x.H, x.W, x.C, _stream_ptr()))
        e0 = tm.start() if tm is not None else None
        N.check(N.lib().gsw_gn_proj_tokens(x.rows.data_ptr(), ws.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), norm.num_groups, w.data_ptr(),
                                           y.data_ptr(), ostat.data_ptr() if ostat is not None else None, float(eps_next) if eps_next is not None else 0.0,
                                           x.B, x.H, x.W, x.C, _dt(dt), _stream_ptr()))
        if tm is not None:
            M = x.B * x.H * x.W
            tm.stop(e0, ("gsw_gnproj_kernel", M, x.C, x.C, "gn+proj_in") if tm.by_shape else "gsw_gnproj_kernel", 2.0 * M * x.C * (x.C + 16),
                    nbytes=2.0 * (x.B * (x.H + 2) * (x.W + 2) + M) * x.C + w.numel() * 2.0)
    if ostat is not None:
        y._gsw_lnstat = (ostat, float(eps_next))
    return y


class _LinearView:
    """a 1 x 1 convolution seen as the linear layer it is (SD 1.5's proj_in): .weight [N, K], .bias"""

    def __init__(self, conv):
        self._conv = conv
        self.weight = conv.weight.view(conv.weight.shape[0], -1)
        self.bias = conv.bias


def _as_linear(lin):
    if lin.weight.dim() == 2:
        return lin
    v = getattr(lin, "_gsw_linear_view", None)
    if v is None or v.weight.data_ptr() != lin.weight.data_ptr():
        v = _LinearView(lin)
        lin._gsw_linear_view = v
    return v
