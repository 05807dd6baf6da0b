"""GPU: the cross-attention sublayer as one launch (gsw_xattn_fused, csrc/gswm_xattn.hip; diffusers' BasicTransformerBlock.attn2 + norm2 + residual behind
extract.py:66-69) against (a) the fp32 torch evaluation of the modules, (b) the three-launch path it replaces (LayerNorm-folded query projection, 77-key
attention kernel, output projection + residual), (c) the lane-level restatement of tests/test_xattn_host.py.  Tolerances are absolute on outputs of scale ~1-3:
fp16 4e-3 x max|ref| (the three-launch path is held to the same bound by tests/test_gpu_lnfold.py), bf16 3e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_xattn_host import _module, emulate, emulate_pre, reference  # noqa: E402


def _stat(x, eps):
    xf = x.float()
    mean, var = xf.mean(-1), xf.var(-1, unbiased=False)
    rstd = torch.rsqrt(var + eps)
    return torch.stack([rstd, -rstd * mean], dim=-1).reshape(-1, 2).contiguous()


def _case(heads, head_dim, ctx_dim, keys, dtype, xB, oB, S, seed, shared_ctx=False):
    from gswm_amd import xattn
    attn, norm = _module(heads, head_dim, ctx_dim, dtype, seed)
    attn, norm = attn.cuda(), norm.cuda()
    g = torch.Generator().manual_seed(seed + 100)
    x = (torch.randn(xB, S, 320, generator=g) * 1.3 + 0.4).to(dtype).cuda()
    if shared_ctx:
        ctx = torch.randn(1, keys, ctx_dim, generator=g).to(dtype).cuda().expand(oB, -1, -1)
    else:
        ctx = torch.randn(oB, keys, ctx_dim, generator=g).to(dtype).cuda()
    blob, uv, idx = xattn.context_operands(attn, norm, ctx, dtype)
    y = xattn.fused(x, _stat(x, norm.eps), blob, uv, idx, oB, heads, eps_out=1e-5)
    want = torch.cat([reference(x, norm, attn, ctx[i * xB:(i + 1) * xB]) for i in range(oB // xB)], dim=0)
    return x, ctx, attn, norm, blob, uv, y, want


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("heads,head_dim,ctx_dim,keys,xB,oB,S,shared", [
    (5, 64, 1024, 77, 2, 2, 256, False),        # SD 2.1 level 0, one context per image
    (5, 64, 1024, 77, 3, 3, 128, True),         # the empty prompt for every image (extract.py:66): one stream
    (5, 64, 1024, 77, 2, 4, 384, False),        # classifier-free guidance on shared latents: 2 x, 4 contexts
    (8, 40, 768, 77, 1, 1, 1152, False),        # SD 1.5 level 0: 8 heads of 40
    (2, 64, 64, 79, 1, 1, 128, False),          # every key slot live
    (3, 32, 96, 5, 2, 2, 128, False),           # almost all padding
])
def test_fused_vs_fp32_reference(heads, head_dim, ctx_dim, keys, xB, oB, S, shared, dtype):
    x, ctx, attn, norm, blob, uv, y, want = _case(heads, head_dim, ctx_dim, keys, dtype, xB, oB, S, seed=heads + keys, shared_ctx=shared)
    torch.cuda.synchronize()
    assert y.shape == want.shape and torch.isfinite(y.float()).all()
    tol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, want.abs().max().item())
    assert (y.float() - want).abs().max().item() <= tol
    # the statistics it leaves for the next LayerNorm are those of the rows it stored
    ostat, eps = y._gsw_lnstat
    ref = _stat(y, eps)
    assert torch.allclose(ostat, ref, rtol=2e-3, atol=2e-3)


def test_fused_equals_the_lane_level_restatement():
    """one wave's 32 rows, bit for bit up to fp32 summation order: the kernel and tests/test_xattn_host.emulate round at the same points"""
    heads, dtype = 5, torch.float16
    x, ctx, attn, norm, blob, uv, y, want = _case(heads, 64, 1024, 77, dtype, 1, 1, 128, seed=11)
    st = _stat(x, norm.eps)
    for blk in (0, 1, 3):
        rows = slice(32 * blk, 32 * blk + 32)
        emu = emulate(x[0, rows].cpu(), st[rows].cpu(), blob[0].cpu(), uv[0].cpu(), heads).to(dtype)
        got = y[0, rows].cpu()
        # identical rounding points; the exponentials (v_exp_f32 vs torch.exp2) and the fp32 summation order differ by ulps of fp32 -> at most one fp16 ulp apart
        diff = (got.float() - emu.float()).abs()
        assert diff.max().item() <= 2.0 ** -9 * max(1.0, emu.float().abs().max().item()) * 2
        assert (diff > 0).float().mean().item() < 0.05


def _pre_case(heads, head_dim, ctx_dim, keys, dtype, xB, oB, S, seed):
    """gsw_xattn_fused_pre: the self-attention's output projection + bias + residual + norm2's statistics as the launch's prologue"""
    from gswm_amd import xattn
    attn, norm = _module(heads, head_dim, ctx_dim, dtype, seed)
    attn, norm = attn.cuda(), norm.cuda()
    g = torch.Generator().manual_seed(seed + 200)
    lin = torch.nn.Linear(320, 320)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(320, 320, generator=g) * 1.2 * 320 ** -0.5)
        lin.bias.copy_(0.2 * torch.randn(320, generator=g))
    lin = lin.to(dtype).cuda()
    resid = (torch.randn(xB, S, 320, generator=g) * 1.3 + 0.4).to(dtype).cuda()
    o = torch.randn(xB, S, 320, generator=g).to(dtype).cuda()
    ctx = torch.randn(oB, keys, ctx_dim, generator=g).to(dtype).cuda()
    blob, uv, idx = xattn.context_operands(attn, norm, ctx, dtype)
    w = xattn.out_projection_operand(lin, dtype)
    y = xattn.fused(resid, None, blob, uv, idx, oB, heads, eps_out=1e-5, pre_o=o, pre_w=w, pre_eps=norm.eps)
    # what the separate launches compute: the projection's output ROUNDED to the storage dtype, then the sublayer on it
    x1 = (resid.float() + F.linear(o.float(), lin.weight.float(), lin.bias.float())).to(dtype)
    want = torch.cat([reference(x1, norm, attn, ctx[i * xB:(i + 1) * xB]) for i in range(oB // xB)], dim=0)
    return resid, o, w, x1, attn, norm, blob, uv, y, want


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("heads,head_dim,ctx_dim,keys,xB,oB,S", [
    (5, 64, 1024, 77, 2, 2, 256),        # SD 2.1 level 0
    (5, 64, 1024, 77, 2, 4, 384),        # classifier-free guidance on shared latents: the prologue runs on the shared rows for both halves
    (8, 40, 768, 77, 1, 1, 1152),        # SD 1.5 level 0
    (3, 32, 96, 5, 3, 3, 128),
])
def test_fused_pre_vs_fp32_reference(heads, head_dim, ctx_dim, keys, xB, oB, S, dtype):
    resid, o, w, x1, attn, norm, blob, uv, y, want = _pre_case(heads, head_dim, ctx_dim, keys, dtype, xB, oB, S, seed=heads + keys)
    torch.cuda.synchronize()
    assert y.shape == want.shape and torch.isfinite(y.float()).all()
    tol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, want.abs().max().item())
    assert (y.float() - want).abs().max().item() <= tol
    ostat, eps = y._gsw_lnstat
    assert torch.allclose(ostat, _stat(y, eps), rtol=2e-3, atol=2e-3)
    # ... and close to the plain kernel fed the rounded stream and its statistics (what the two-launch path hands it)
    from gswm_amd import xattn
    idx = torch.arange(oB, dtype=torch.int32, device="cuda") if oB > 1 else None
    y2 = xattn.fused(x1, _stat(x1, norm.eps), blob, uv, idx, oB, heads, eps_out=1e-5)
    assert (y.float() - y2.float()).abs().max().item() <= tol


def test_fused_pre_equals_the_lane_level_restatement():
    """prologue + sublayer for one wave's 32 rows against tests/test_xattn_host.emulate_pre -> emulate: same rounding points"""
    heads, dtype = 5, torch.float16
    resid, o, w, x1, attn, norm, blob, uv, y, want = _pre_case(heads, 64, 1024, 77, dtype, 1, 1, 256, seed=13)
    for blk in (0, 2, 7):
        rows = slice(32 * blk, 32 * blk + 32)
        x, st = emulate_pre(o[0, rows].cpu(), resid[0, rows].cpu(), w.cpu(), norm.eps)
        emu = emulate(x, st, blob[0].cpu(), uv[0].cpu(), heads).to(dtype)
        got = y[0, rows].cpu()
        diff = (got.float() - emu.float()).abs()
        assert diff.max().item() <= 2.0 ** -9 * max(1.0, emu.float().abs().max().item()) * 2
        assert (diff > 0).float().mean().item() < 0.05


def test_fused_pre_full_chip():
    """more tiles than workgroups: every workgroup's stream wraps from the last head of a tile into the next tile's prologue"""
    resid, o, w, x1, attn, norm, blob, uv, y, want = _pre_case(5, 64, 1024, 77, torch.float16, 16, 16, 4096, seed=3)
    tol = 4e-3 * max(1.0, want.abs().max().item())
    assert (y.float() - want).abs().max().item() <= tol


@pytest.mark.parametrize("B", [16, 21])
def test_fused_full_chip_xcd_order(B):
    """more tiles than workgroups, >= 8 images: the XCD-ordered tile walk (images distributed over the 8 L2s) incl. an image count that is not a multiple of 8"""
    heads, dtype, S = 5, torch.float16, 4096
    x, ctx, attn, norm, blob, uv, y, want = _case(heads, 64, 1024, 77, dtype, B, B, S, seed=B)
    tol = 4e-3 * max(1.0, want.abs().max().item())
    assert (y.float() - want).abs().max().item() <= tol


def test_block_one_launch_vs_three_launches():
    """the whole BasicTransformerBlock with and without the one-launch cross-attention: same bound against fp32 torch, and close to each other"""
    from gswm_amd import unet as U, xattn, pf
    torch.manual_seed(0)
    blk = U.BasicTransformerBlock(320, 1024, 5, 64)
    for p_ in blk.parameters():
        if p_.dim() == 2:
            torch.nn.init.normal_(p_, std=1.2 * p_.shape[1] ** -0.5)
    blk = blk.cuda().half().eval()
    B, S = 2, 1024
    x0 = torch.randn(B, S, 320, device="cuda").half()
    ctx = torch.randn(B, 77, 1024, device="cuda").half()
    w = (torch.randn(320, 320, device="cuda") * 320 ** -0.5).half()

    def run(flag, pre=True):
        xattn.ENABLED, xattn.PRE_ENABLED = flag, pre
        with torch.no_grad():
            x = pf.gemm(x0, w, None, rowstats=True)            # a producer that leaves row records, like proj_in
            return blk(x, ctx), x
    try:
        y1, x = run(True)
        y1b, _ = run(True, pre=False)                           # one launch for the cross-attention, the self-attention's projection on its own
        y3, _ = run(False)
    finally:
        xattn.ENABLED = xattn.PRE_ENABLED = True
    with torch.no_grad():
        f = blk.float()
        xf = x.float()
        h = xf + f.attn1(f.norm1(xf))
        h = h + f.attn2(f.norm2(h), ctx.float())
        ref = h + f.ff(f.norm3(h))
        blk.half()
    scale = max(1.0, ref.abs().max().item())
    assert (y1.float() - ref).abs().max().item() <= 1e-2 * scale
    assert (y3.float() - ref).abs().max().item() <= 1e-2 * scale
    assert (y1.float() - y3.float()).abs().max().item() <= 1e-2 * scale
    assert (y1b.float() - ref).abs().max().item() <= 1e-2 * scale


def test_unsupported_shapes_are_refused_not_computed():
    from gswm_amd import xattn, _native
    x = torch.zeros(1, 100, 320, device="cuda").half()
    with pytest.raises(ValueError):
        xattn.fused(x, torch.zeros(100, 2, device="cuda"), torch.zeros(1, 5 * xattn.HEAD_ELEMS, device="cuda").half(), torch.zeros(1, 5 * xattn.V_FLOATS, device="cuda"), None, 1, 5)
    lib = _native.lib()
    z = torch.zeros(4096, device="cuda")
    assert lib.gsw_xattn_fused(z.data_ptr(), z.data_ptr(), z.data_ptr(), 16, z.data_ptr(), 4, None, z.data_ptr(), None, 0.0, 1, 1, 128, 640, 5, 1, None) == _native.GSW_ERR_UNSUPPORTED
    assert lib.gsw_xattn_fused(z.data_ptr(), z.data_ptr(), z.data_ptr(), 16, z.data_ptr(), 4, None, z.data_ptr(), None, 0.0, 1, 1, 100, 320, 5, 1, None) == _native.GSW_ERR_UNSUPPORTED
    assert lib.gsw_xattn_fused(None, z.data_ptr(), z.data_ptr(), 16, z.data_ptr(), 4, None, z.data_ptr(), None, 0.0, 1, 1, 128, 320, 5, 1, None) == _native.GSW_ERR_BAD_ARG


def test_unet_forward_with_and_without_the_one_launch_cross_attention():
    """the whole SD 2.1-shaped UNet, 64 x 64 latents, plain and classifier-free-guidance (shared latents) forwards: the one-launch cross-attention of the five
    320-channel blocks against the three launches it replaces -- same eps to fp16 noise level, and nothing leaves the hand-written path either way"""
    from gswm_amd import unet as U, xattn
    torch.manual_seed(0)
    m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
    x = torch.randn(2, 4, 64, 64, device="cuda").half()
    t = torch.full((), 481, device="cuda")
    ctx = torch.randn(2, 77, 1024, device="cuda").half()
    ctx2 = torch.cat([torch.randn(1, 77, 1024, device="cuda").half().expand(2, -1, -1), ctx], dim=0)      # (uncond x 2 | text x 2)
    U.FALLBACKS.clear()
    outs = {}
    try:
        for flag in (True, "no-pre", False):
            xattn.ENABLED, xattn.PRE_ENABLED = bool(flag), flag is True
            with torch.no_grad():
                outs[flag] = (m(x, t, ctx).float(), m(x, t, ctx2, cfg_dup=True).float())
    finally:
        xattn.ENABLED = xattn.PRE_ENABLED = True
    assert U.FALLBACKS == {}
    for other in (False, "no-pre"):
        for a, b in zip(outs[True], outs[other]):
            assert a.shape == b.shape and torch.isfinite(a).all()
            assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item())
    # the guidance batch of shared latents equals the doubled batch (what the reference computes: torch.cat([latents] * 2))
    with torch.no_grad():
        doubled = m(torch.cat([x, x], dim=0), t, ctx2).float()
    assert (outs[True][1] - doubled).abs().max().item() <= 2e-2 * max(1.0, doubled.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("B,H,W", [(3, 64, 64), (2, 96, 96), (5, 32, 32), (1, 16, 32)])
def test_gn_proj_tokens_vs_fp32_reference(B, H, W, dtype):
    """GroupNorm + proj_in of a transformer as one launch (gsw_gn_proj_tokens) on a PF tensor whose producing convolution left the column records: against the fp32
    modules on the SAME stored tensor, against the two launches it replaces (gsw_gn_pf_apply -> tokens, engine GEMM), and the (rstd, -rstd mean) it leaves"""
    from gswm_amd import pf, xattn
    g = torch.Generator().manual_seed(B * H + W)
    src = pf.PF.from_nchw((torch.randn(B, 64, H, W, generator=g) * 1.5).to(dtype).cuda())
    cw = pf.pack_conv_weight((torch.randn(320, 64, 3, 3, generator=g) * 0.06).to(dtype).cuda())
    cb = (torch.randn(320, generator=g) * 0.5).to(dtype).cuda()               # channel means far from zero: the shift term matters
    x = pf.conv_pf(src, cw, cb)
    norm = torch.nn.GroupNorm(32, 320, eps=1e-6)
    lin = torch.nn.Linear(320, 320)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.3 * torch.randn(320, generator=g)); norm.bias.copy_(0.2 * torch.randn(320, generator=g))
        lin.weight.copy_(torch.randn(320, 320, generator=g) * 1.2 * 320 ** -0.5); lin.bias.copy_(0.2 * torch.randn(320, generator=g))
    norm, lin = norm.to(dtype).cuda(), lin.to(dtype).cuda()
    if not xattn.gn_proj_usable(x, norm, lin):
        assert pf._gn_fused_ok(B, H, W, 320, 32) or x.stats is None      # (small grids keep the one-launch GroupNorm of section 4.7)
        pytest.skip("this geometry runs the small-batch GroupNorm")
    y = xattn.gn_proj(x, norm, lin, eps_next=1e-5)
    torch.cuda.synchronize()
    xi = x.interior.float().permute(0, 3, 1, 2)                                # what is stored, NCHW
    ref = F.linear(F.group_norm(xi, 32, norm.weight.float(), norm.bias.float(), norm.eps).permute(0, 2, 3, 1).reshape(B, H * W, 320), lin.weight.float(), lin.bias.float())
    assert y.shape == ref.shape and torch.isfinite(y.float()).all()
    tol = (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, ref.abs().max().item())
    assert (y.float() - ref).abs().max().item() <= tol
    two = pf.gemm(pf.groupnorm_pf(x, norm.weight, norm.bias, 32, norm.eps, act=False, tokens=True), lin.weight, lin.bias)
    assert (y.float() - two.float()).abs().max().item() <= tol
    ostat, eps = y._gsw_lnstat
    assert torch.allclose(ostat, _stat(y, eps), rtol=2e-3, atol=2e-3)


def test_unet_forward_with_and_without_the_fused_gn_proj():
    """whole SD 2.1-shaped UNet at 16 images (the 64 x 64 transformers take the one-launch GroupNorm + proj_in, the others keep the two launches): same eps"""
    from gswm_amd import unet as U, xattn
    torch.manual_seed(0)
    m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
    x = torch.randn(16, 4, 64, 64, device="cuda").half()
    t = torch.full((), 481, device="cuda")
    ctx = torch.randn(16, 77, 1024, device="cuda").half()
    U.FALLBACKS.clear()
    outs = {}
    try:
        for flag in (True, False):
            xattn.GNPROJ_ENABLED = flag
            with torch.no_grad():
                outs[flag] = m(x, t, ctx).float()
    finally:
        xattn.GNPROJ_ENABLED = True
    assert U.FALLBACKS == {}
    assert torch.isfinite(outs[True]).all()
    assert (outs[True] - outs[False]).abs().max().item() <= 2e-2 * max(1.0, outs[False].abs().max().item())
