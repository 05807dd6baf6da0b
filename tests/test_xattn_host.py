"""CPU: the host side of the one-launch cross-attention sublayer (xattn.py -> csrc/gswm_xattn.hip; diffusers' BasicTransformerBlock.attn2 behind
extract.py:66-69).  `emulate` below restates, in plain torch and straight from the BYTES of the fragment stream, what the kernel computes lane by lane
(v_mfma_f32_32x32x16 operand / accumulator layouts, chunk order, key-slot and column permutations, the bias key, the LayerNorm fold) and is held to the
fp32 evaluation of the torch modules; the GPU tests (tests/test_gpu_xattn.py) hold the kernel to both.  This is test infrastructure: the product path has
no CPU twin."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import gswm_amd  # noqa: F401
from gswm_amd import xattn
from gswm_amd.unet import Attention


def mfma_32x32x16(a_frag: torch.Tensor, b_frag: torch.Tensor) -> torch.Tensor:
    """v_mfma_f32_32x32x16: a_frag, b_frag [64 lanes, 8] (lane = row-or-column + 32 hlf, value e <-> k = 8 hlf + e) -> accumulator [64 lanes, 16]: lane (col, hlf)
    element t = D[8 (t >> 2) + 4 hlf + (t & 3)][col]"""
    A = torch.zeros(32, 16)
    Bm = torch.zeros(16, 32)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        A[r, 8 * h:8 * h + 8] = a_frag[lane]
        Bm[8 * h:8 * h + 8, r] = b_frag[lane]
    D = A @ Bm
    out = torch.zeros(64, 16)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        for t in range(16):
            out[lane, t] = D[8 * (t >> 2) + 4 * h + (t & 3), r]
    return out


def emulate(x: torch.Tensor, stat: torch.Tensor, blob: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """One wave's 32 rows of x [32, 320] through the kernel's instruction-level algebra -> x' [32, 320] (fp32, before the final rounding)."""
    dt = x.dtype
    xf = [torch.stack([x[lane & 31, 16 * ks + 8 * (lane >> 5):16 * ks + 8 * (lane >> 5) + 8] for lane in range(64)]).float() for ks in range(20)]
    frags = blob.view(heads, 110, 64, 8).float()
    vv = v.view(heads, 96)
    # residual through the matrix pipe: permutation fragments
    pm = []
    for j in range(2):
        a = torch.zeros(64, 8)
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            if (r >> 4) == j and ((r >> 2) & 1) == h:
                a[lane, 4 * ((r >> 3) & 1) + (r & 3)] = 1.0
        pm.append(a)
    acc = [mfma_32x32x16(pm[0], xf[2 * nb]) + mfma_32x32x16(pm[1], xf[2 * nb + 1]) for nb in range(10)]
    for h in range(heads):
        S = [torch.zeros(64, 16) for _ in range(3)]
        for f in range(60):
            ks, kb = f // 3, f % 3
            S[kb] = S[kb] + mfma_32x32x16(frags[h, f], xf[ks])
        s = torch.zeros(64, 40)
        for lane in range(64):
            r, hl = lane & 31, lane >> 5
            for q in range(10):
                kb, tq = q >> 2, q & 3
                for i in range(4):
                    key = 32 * kb + 8 * tq + 4 * hl + i
                    s[lane, 4 * q + i] = stat[r, 0] * S[kb][lane, 4 * tq + i] + vv[h, key]
        P = torch.zeros(64, 40)
        for r in range(32):
            lanes = [r, r + 32]
            m = s[lanes].max()
            e = torch.exp2(s[lanes] - m)
            P[lanes] = e / e.sum()
        for kk in range(5):
            pf = P[:, 8 * kk:8 * kk + 8].clone()
            if kk == 4:
                pf[32:, 7] = 1.0                 # key slot 79: the bias row
            pf = pf.to(dt).float()
            for nb in range(10):
                acc[nb] = acc[nb] + mfma_32x32x16(frags[h, 60 + kk * 10 + nb], pf)
    out = torch.zeros(32, 320)
    for lane in range(64):
        r, hl = lane & 31, lane >> 5
        for nb in range(10):
            for j in range(2):
                out[r, 32 * nb + 16 * j + 8 * hl:32 * nb + 16 * j + 8 * hl + 8] = acc[nb][lane, 8 * j:8 * j + 8]
    return out


def emulate_pre(o: torch.Tensor, resid: torch.Tensor, wfrag: torch.Tensor, eps: float):
    """gsw_xattn_fused_pre's prologue for one wave's 32 rows, from the BYTES of the projection's fragment stream: x = resid + o Wo^T + b through the same MFMA layouts
    (twenty k-steps of ten column blocks, the bias against a unit vector, the residual against the permutation fragments), rounded once -> (x [32, 320] in the storage
    dtype, its (rstd, -rstd mean) [32, 2] taken from the rounded values)"""
    dt = o.dtype
    frag_of = lambda t: [torch.stack([t[lane & 31, 16 * ks + 8 * (lane >> 5):16 * ks + 8 * (lane >> 5) + 8] for lane in range(64)]).float() for ks in range(20)]
    of, rf = frag_of(o), frag_of(resid)
    fr = wfrag.view(xattn.PRE_CHUNKS, 10, 64, 8).float()
    e0 = torch.zeros(64, 8)
    e0[:32, 0] = 1.0
    pm = []
    for j in range(2):
        a = torch.zeros(64, 8)
        for lane in range(64):
            r, h = lane & 31, lane >> 5
            if (r >> 4) == j and ((r >> 2) & 1) == h:
                a[lane, 4 * ((r >> 3) & 1) + (r & 3)] = 1.0
        pm.append(a)
    acc = [torch.zeros(64, 16) for _ in range(10)]
    for ks in range(20):
        for nb in range(10):
            acc[nb] = acc[nb] + mfma_32x32x16(fr[ks, nb], of[ks])
    for nb in range(10):
        acc[nb] = acc[nb] + mfma_32x32x16(fr[20, nb], e0)
        acc[nb] = acc[nb] + mfma_32x32x16(pm[0], rf[2 * nb]) + mfma_32x32x16(pm[1], rf[2 * nb + 1])
    x = torch.zeros(32, 320)
    for lane in range(64):
        r, hl = lane & 31, lane >> 5
        for nb in range(10):
            for j in range(2):
                x[r, 32 * nb + 16 * j + 8 * hl:32 * nb + 16 * j + 8 * hl + 8] = acc[nb][lane, 8 * j:8 * j + 8]
    x = x.to(dt)
    xf = x.float()
    mean = xf.mean(-1)
    var = ((xf * xf).mean(-1) - mean * mean).clamp_min(0)
    rstd = torch.rsqrt(var + eps)
    return x, torch.stack([rstd, -rstd * mean], dim=1)


def reference(x, norm, attn, ctx):
    """fp32 torch: x + to_out(attention(to_q(LayerNorm(x)), to_k(ctx), to_v(ctx)))"""
    f = torch.float32
    n = F.layer_norm(x.to(f), (x.shape[-1],), norm.weight.to(f), norm.bias.to(f), norm.eps)
    H = attn.heads
    q = F.linear(n, attn.to_q.weight.to(f))
    k = F.linear(ctx.to(f), attn.to_k.weight.to(f))
    v = F.linear(ctx.to(f), attn.to_v.weight.to(f))
    split = lambda t: t.view(t.shape[0], t.shape[1], H, -1).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(q), split(k), split(v)).transpose(1, 2).reshape(x.shape[0], x.shape[1], -1)
    return x.to(f) + F.linear(o, attn.to_out[0].weight.to(f), attn.to_out[0].bias.to(f))


def _module(heads, head_dim, ctx_dim, dtype, seed):
    torch.manual_seed(seed)
    attn = Attention(320, ctx_dim, heads, head_dim)
    norm = nn.LayerNorm(320)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.2 * torch.randn(320))
        norm.bias.copy_(0.1 * torch.randn(320))
        attn.to_out[0].bias.copy_(0.1 * torch.randn(320))
        for lin in (attn.to_q, attn.to_k, attn.to_v, attn.to_out[0]):
            lin.weight.copy_(torch.randn_like(lin.weight) * 1.5 * lin.in_features ** -0.5)
    return attn.to(dtype), norm.to(dtype)


@pytest.mark.parametrize("heads,head_dim,ctx_dim,keys", [(5, 64, 1024, 77), (8, 40, 768, 77), (2, 64, 64, 79), (3, 32, 96, 5)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_fragment_stream_restates_the_sublayer(heads, head_dim, ctx_dim, keys, dtype):
    attn, norm = _module(heads, head_dim, ctx_dim, dtype, seed=heads)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(1, 32, 320, generator=g) * 1.3 + 0.4).to(dtype)
    ctx = torch.randn(1, keys, ctx_dim, generator=g).to(dtype)
    Ap, v, Bm = xattn.fold_operands(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias, norm.weight, norm.bias,
                                    ctx, heads, dtype)
    assert Ap.shape == (1, heads, 96, 320) and Bm.shape == (1, heads, 320, 80)
    assert torch.isinf(v[0, :, keys:]).all() and (v[0, :, keys:] < 0).all() and (Ap[0, :, keys:] == 0).all()
    # centred rows: the LayerNorm mean needs no rank-one term; what the rounding leaves of a row sum (times mean / std of a token) is the error this costs the scores
    assert Ap[0, :, :keys].float().sum(-1).abs().max().item() <= (2e-2 if dtype == torch.float16 else 2e-1)
    assert (Bm[0, :-1, :, 79] == 0).all() and torch.equal(Bm[0, -1, :, 79], attn.to_out[0].bias.detach())
    blob, vf = xattn.pack_stream(Ap, v, Bm)
    assert blob.shape == (1, heads * xattn.HEAD_ELEMS) and blob.dtype == dtype and vf.shape == (1, heads * xattn.V_FLOATS)
    xf = x[0].float()
    mean, var = xf.mean(-1), xf.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + norm.eps)
    stat = torch.stack([rstd, -rstd * mean], dim=1)
    got = emulate(x[0], stat, blob[0], vf[0], heads)
    want = reference(x, norm, attn, ctx)[0]
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), (got - want).abs().max().item()


def test_stream_is_a_permutation_of_the_operands():
    """every element of A' and of B's 80 key slots appears exactly once in the stream (nothing dropped, nothing duplicated)"""
    heads = 2
    Ap = torch.arange(heads * 96 * 320, dtype=torch.float32).view(1, heads, 96, 320) + 1.0
    Bm = -(torch.arange(heads * 320 * 80, dtype=torch.float32).view(1, heads, 320, 80) + 1.0)
    blob, _ = xattn.pack_stream(Ap, torch.zeros(1, heads, 96), Bm)
    b = blob.view(heads, xattn.HEAD_ELEMS)
    for h in range(heads):
        g1, g2 = b[h, :30720], b[h, 30720:]
        assert torch.equal(g1.sort().values, Ap[0, h].flatten().sort().values)
        assert torch.equal(g2.sort().values, Bm[0, h].flatten().sort().values)


def test_run_index_groups_identical_neighbours():
    a, b, c = torch.randn(3, 77, 16).unbind(0)
    ctx = torch.stack([a, a, a, b, c, c, a])
    assert xattn.run_index(ctx).tolist() == [0, 0, 0, 3, 4, 4, 6]
    assert xattn.run_index(ctx[:1]) is None


def test_expanded_context_is_one_stream_and_cache_follows_versions():
    attn, norm = _module(5, 64, 32, torch.float16, seed=3)
    base = torch.randn(1, 77, 32).half()
    ctx = base.expand(6, -1, -1)
    blob, uv, idx = xattn.context_operands(attn, norm, ctx, torch.float16)
    assert blob.shape[0] == 1 and idx is None
    again = xattn.context_operands(attn, norm, ctx, torch.float16)
    assert again[0] is blob and again[1] is uv
    ptr = blob.data_ptr()
    with torch.no_grad():
        norm.weight.mul_(1.5)                    # a parameter edit: recomputed INTO the same buffers
    blob2, uv2, _ = xattn.context_operands(attn, norm, ctx, torch.float16)
    assert blob2.data_ptr() == ptr and not torch.equal(uv2, torch.zeros_like(uv2))


def test_usable_gates():
    attn, _ = _module(5, 64, 32, torch.float16, seed=4)
    x = torch.empty(2, 256, 320, dtype=torch.float16)
    ctx = torch.empty(2, 77, 32, dtype=torch.float16)
    assert not xattn.usable(x, attn, ctx)          # CPU tensors never qualify: no CPU fallback
    with pytest.raises(RuntimeError):
        xattn.fused(x, torch.empty(512, 2), torch.empty(1, 5 * xattn.HEAD_ELEMS, dtype=torch.float16), torch.empty(1, 5 * xattn.V_FLOATS), None, 2, 5)
    assert math.isclose(xattn.LOG2E, math.log2(math.e))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_out_projection_stream_restates_the_projection(dtype):
    """gsw_xattn_fused_pre's prologue operand: the fragment stream of the self-attention's output projection, evaluated lane by lane, is resid + o Wo^T + b"""
    torch.manual_seed(7)
    lin = nn.Linear(320, 320)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(320, 320) * 1.3 * 320 ** -0.5)
        lin.bias.copy_(0.2 * torch.randn(320))
    lin = lin.to(dtype)
    o = torch.randn(32, 320).to(dtype)
    resid = (torch.randn(32, 320) * 1.2 + 0.3).to(dtype)
    w = xattn.pack_out_projection(lin.weight, lin.bias, dtype)
    assert w.shape == (xattn.PRE_CHUNKS * 5120,) and w.dtype == dtype
    # nothing dropped, nothing duplicated: the first twenty chunks are a permutation of Wo, the last holds b and zeros
    assert torch.equal(w[:20 * 5120].float().sort().values, lin.weight.detach().float().flatten().sort().values)
    assert torch.equal(w[20 * 5120:].float().sort().values, torch.cat([lin.bias.detach().float(), torch.zeros(5120 - 320)]).sort().values)
    x, stat = emulate_pre(o, resid, w, 1e-5)
    want = resid.float() + F.linear(o.float(), lin.weight.float(), lin.bias.float())
    tol = (2e-3 if dtype == torch.float16 else 1.6e-2) * max(1.0, want.abs().max().item())
    assert (x.float() - want).abs().max().item() <= tol
    ref = torch.stack([torch.rsqrt(want.var(-1, unbiased=False) + 1e-5), -torch.rsqrt(want.var(-1, unbiased=False) + 1e-5) * want.mean(-1)], dim=1)
    assert torch.allclose(stat, ref, rtol=1e-2, atol=1e-2)
    # cached on the module; a parameter edit recomputes into the same buffer
    a = xattn.out_projection_operand(lin, dtype)
    assert xattn.out_projection_operand(lin, dtype) is a and torch.equal(a, w)
    ptr = a.data_ptr()
    with torch.no_grad():
        lin.bias.add_(1.0)
    b = xattn.out_projection_operand(lin, dtype)
    assert b.data_ptr() == ptr and not torch.equal(b, w)


def test_gn_proj_host_gates_and_the_linear_view_of_a_1x1_convolution():
    """gsw_gn_proj_tokens' host side: nothing qualifies on the CPU (no fallback), and SD 1.5's 1 x 1-convolution proj_in is packed as the linear layer it is"""
    from gswm_amd import pf
    x = pf.PF.from_nchw(torch.randn(1, 320, 8, 32).half())
    norm = nn.GroupNorm(32, 320).half()
    lin = nn.Linear(320, 320).half()
    assert not xattn.gn_proj_usable(x, norm, lin)
    with pytest.raises(RuntimeError):
        xattn.gn_proj(x, norm, lin)
    conv = nn.Conv2d(320, 320, 1).half()
    view = xattn._as_linear(conv)
    assert view.weight.shape == (320, 320) and view.weight.data_ptr() == conv.weight.data_ptr() and xattn._as_linear(conv) is view
    assert torch.equal(xattn.pack_out_projection(view.weight, view.bias, torch.float16),
                       xattn.pack_out_projection(conv.weight.detach().reshape(320, 320), conv.bias, torch.float16))
    assert xattn._as_linear(lin) is lin
