"""CPU: the host side of the one-launch cross-attention sublayer (xattn.py -> csrc/gswm_xattn.hip; diffusers' BasicTransformerBlock.attn2 behind
extract.py:66-69).  `emulate` below restates, in plain torch and straight from the BYTES of the fragment stream, what the kernel computes lane by lane
(v_mfma_f32_16x16x32 operand / accumulator layouts, chunk order, key-slot and column permutations, the bias key, the LayerNorm fold) and is held to the
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


def mfma_16x16x32(a_frag: torch.Tensor, b_frag: torch.Tensor) -> torch.Tensor:
    """a_frag, b_frag [64 lanes, 8] (lane = row-or-column + 16 g, element e <-> k = 8 g + e) -> accumulator [64 lanes, 4]: lane (col, g) element i = D[4 g + i][col]"""
    A = torch.zeros(16, 32)
    Bm = torch.zeros(32, 16)
    for lane in range(64):
        r, g = lane & 15, lane >> 4
        A[r, 8 * g:8 * g + 8] = a_frag[lane]
        Bm[8 * g:8 * g + 8, r] = b_frag[lane]
    D = A @ Bm
    out = torch.zeros(64, 4)
    for lane in range(64):
        r, g = lane & 15, lane >> 4
        out[lane] = D[4 * g:4 * g + 4, r]
    return out


def emulate(x: torch.Tensor, stat: torch.Tensor, blob: torch.Tensor, uv: torch.Tensor, heads: int) -> torch.Tensor:
    """One 16-row block of x [16, 320] through the kernel's instruction-level algebra -> x' [16, 320] (fp32, before the final rounding)."""
    dt = x.dtype
    xf = [torch.stack([x[lane & 15, 32 * ks + 8 * (lane >> 4):32 * ks + 8 * (lane >> 4) + 8] for lane in range(64)]).float() for ks in range(10)]
    frags = blob.view(heads, 110, 64, 8).float()
    uvv = uv.view(heads, 2, 80)
    # residual through the matrix pipe: permutation fragments
    acc = []
    for nb in range(20):
        pm = torch.zeros(64, 8)
        for lane in range(64):
            r, g = lane & 15, lane >> 4
            if g == (r >> 2):
                pm[lane, 4 * (nb & 1) + (r & 3)] = 1.0
        acc.append(mfma_16x16x32(pm, xf[nb >> 1]))
    for h in range(heads):
        S = [torch.zeros(64, 4) for _ in range(5)]
        for ks in range(10):
            for kb in range(5):
                S[kb] = S[kb] + mfma_16x16x32(frags[h, ks * 5 + kb], xf[ks])
        s = torch.zeros(64, 20)
        for lane in range(64):
            r, g = lane & 15, lane >> 4
            for kb in range(5):
                for i in range(4):
                    key = 16 * kb + 4 * g + i
                    s[lane, 4 * kb + i] = stat[r, 0] * S[kb][lane, i] + (stat[r, 1] * uvv[h, 0, key] + uvv[h, 1, key])
        P = torch.zeros(64, 20)
        for r in range(16):
            lanes = [r + 16 * g for g in range(4)]
            m = s[lanes].max()
            e = torch.exp2(s[lanes] - m)
            P[lanes] = e / e.sum()
        one = torch.tensor(1.0)
        for kk in range(3):
            pf = torch.zeros(64, 8)
            for lane in range(64):
                g = lane >> 4
                pf[lane, 0:4] = P[lane, 8 * kk:8 * kk + 4]
                if kk < 2:
                    pf[lane, 4:8] = P[lane, 8 * kk + 4:8 * kk + 8]
                elif g == 0:
                    pf[lane, 4] = one
            pf = pf.to(dt).float()
            for nb in range(20):
                acc[nb] = acc[nb] + mfma_16x16x32(frags[h, 50 + kk * 20 + nb], pf)
    out = torch.zeros(16, 320)
    for lane in range(64):
        r, g = lane & 15, lane >> 4
        for q in range(10):
            out[r, 32 * q + 8 * g:32 * q + 8 * g + 4] = acc[2 * q][lane]
            out[r, 32 * q + 8 * g + 4:32 * q + 8 * g + 8] = acc[2 * q + 1][lane]
    return out


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


@pytest.mark.parametrize("heads,head_dim,ctx_dim,keys", [(5, 64, 1024, 77), (8, 40, 768, 77), (2, 64, 64, 80), (3, 32, 96, 5)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_fragment_stream_restates_the_sublayer(heads, head_dim, ctx_dim, keys, dtype):
    attn, norm = _module(heads, head_dim, ctx_dim, dtype, seed=heads)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(1, 16, 320, generator=g) * 1.3 + 0.4).to(dtype)
    ctx = torch.randn(1, keys, ctx_dim, generator=g).to(dtype)
    Ap, u, v, Bm = xattn.fold_operands(attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight, attn.to_out[0].bias, norm.weight, norm.bias,
                                       ctx, heads, dtype)
    assert Ap.shape == (1, heads, 80, 320) and Bm.shape == (1, heads, 320, 96)
    assert torch.isinf(v[0, :, keys:]).all() and (v[0, :, keys:] < 0).all() and (u[0, :, keys:] == 0).all() and (Ap[0, :, keys:] == 0).all()
    assert (Bm[0, :-1, :, 80] == 0).all() and torch.equal(Bm[0, -1, :, 80], attn.to_out[0].bias.detach())
    blob, uv = xattn.pack_stream(Ap, u, v, Bm)
    assert blob.shape == (1, heads * xattn.HEAD_ELEMS) and blob.dtype == dtype and uv.shape == (1, heads * xattn.UV_FLOATS)
    xf = x[0].float()
    mean, var = xf.mean(-1), xf.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + norm.eps)
    stat = torch.stack([rstd, -rstd * mean], dim=1)
    got = emulate(x[0], stat, blob[0], uv[0], heads)
    want = reference(x, norm, attn, ctx)[0]
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item()), (got - want).abs().max().item()


def test_stream_is_a_permutation_of_the_operands():
    """every element of A' and of B's 81 live key slots appears exactly once in the stream (nothing dropped, nothing duplicated)"""
    heads = 2
    Ap = torch.arange(heads * 80 * 320, dtype=torch.float32).view(1, heads, 80, 320) + 1.0
    Bm = -(torch.arange(heads * 320 * 96, dtype=torch.float32).view(1, heads, 320, 96) + 1.0)
    u = torch.zeros(1, heads, 80)
    blob, uv = xattn.pack_stream(Ap, u, u, Bm)
    b = blob.view(heads, xattn.HEAD_ELEMS)
    for h in range(heads):
        g1, g2 = b[h, :25600], b[h, 25600:]
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
        xattn.fused(x, torch.empty(512, 2), torch.empty(1, 5 * xattn.HEAD_ELEMS, dtype=torch.float16), torch.empty(1, 5 * xattn.UV_FLOATS), None, 2, 5)
    assert math.isclose(xattn.LOG2E, math.log2(math.e))
