"""GPU: LayerNorm folded into the GEMMs that consume it (GswMmExtras.rowstats_* -> gsw_ln_rowstats_finish -> gsw_gemm_ln; diffusers BasicTransformerBlock's
norm1 / norm2 / norm3 in front of to_q | to_k | to_v, to_q and the GEGLU projection, behind extract.py:66-69) against the LayerNorm kernel + plain GEMM it
replaces and against fp32 torch: the row records themselves, every consumer mode, both tile heights, ragged N, a DC offset on the residual stream (the
cancellation rstd (x W') - rstd mean u), and the whole transformer block."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, unet, codec, _native
    old = pf.SMALL_GEMM_MAX_ROWS
    pf.SMALL_GEMM_MAX_ROWS = 0          # this module tests the matmul ENGINE (at <= 128 rows pf.gemm would otherwise take gsw_gemm_small: tests/test_gpu_small.py)
    yield types.SimpleNamespace(pf=pf, unet=unet, codec=codec, lib=_native.lib())
    pf.SMALL_GEMM_MAX_ROWS = old


@pytest.fixture(params=[0, 128, 256, 512], ids=["auto", "BM128", "BM256", "WIDE256x320"])      # 512: the 256 x 320 tile wherever it is legal (M % 256 == 0, N % 320 == 0)
def tile_rows(request, G):
    assert G.lib.gsw_mm_config(request.param, -1) == 0
    prev = G.pf.FOLD_LN_MIN_ROWS
    G.pf.FOLD_LN_MIN_ROWS = 0
    yield request.param
    G.pf.FOLD_LN_MIN_ROWS = prev
    assert G.lib.gsw_mm_config(0, -1) == 0


def _producer(G, M, K, N, seed, offset=0.0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().cuda()
    b = (torch.randn(N, generator=g) + offset).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda()
    return G.pf.gemm(a, w, b, resid=r, rowstats=True)


@pytest.mark.parametrize("M,C", [(1024, 320), (520, 640), (4096, 1280), (264, 328)])
def test_row_records_and_stat(G, tile_rows, M, C):
    x = _producer(G, M, 320, C, M + C, offset=1.5)
    rs = getattr(x, "_gsw_rowstats", None)
    assert rs is not None and rs[1] == 2 * ((C + 159) // 160)
    rec = rs[0].view(M, rs[1], 2).double()
    xf = x.double()
    assert torch.allclose(rec[..., 0].sum(1), xf.sum(1), rtol=1e-5, atol=1e-2)
    assert torch.allclose(rec[..., 1].sum(1), (xf * xf).sum(1), rtol=1e-5, atol=1e-2)
    st = G.pf.ln_stat(x, 1e-5)
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    rstd = (var + 1e-5).rsqrt()
    assert torch.allclose(st[:, 0].double(), rstd, rtol=2e-4) and torch.allclose(st[:, 1].double(), -rstd * mean, rtol=2e-4, atol=1e-4)


@pytest.mark.parametrize("mode", ["plain", "geglu", "trans"])
# (40960 rows: several tiles per workgroup -- the staged epilogue parameters of consecutive tiles share one LDS slot; K = 128: two steps per tile, staged inside the epilogue)
@pytest.mark.parametrize("M,C,N", [(2048, 320, 640), (1024, 640, 1920), (512, 1280, 1280), (40960, 320, 1920), (1024, 128, 320)])
def test_folded_gemm_vs_layernorm_then_gemm_vs_fp32(G, tile_rows, mode, M, C, N):
    if mode == "geglu":
        N = (N // 160) * 160
    x = _producer(G, M, 320, C, M + C + N, offset=2.0)              # rows with a DC offset of ~2 sigma
    g = torch.Generator().manual_seed(N)
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).half().cuda()
    beta = (0.2 * torch.randn(C, generator=g)).half().cuda()
    w = (torch.randn(N, C, generator=g) * C ** -0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    st = G.pf.ln_stat(x, 1e-5)
    assert st is not None
    f = G.pf.fold_ln_weights(w, b, gamma, beta, geglu=mode == "geglu")
    n32 = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    proj = n32 @ w.float().T + b.float()
    _, n16 = G.codec.add_layernorm(x, None, gamma, beta, 1e-5)
    if mode == "plain":
        y = G.pf.gemm_ln(x, st, *f)
        ref, old = proj, G.pf.gemm(n16, w, b)
    elif mode == "geglu":
        y = G.pf.gemm_ln(x, st, *f, mode="geglu")
        ref = proj[:, : N // 2] * F.gelu(proj[:, N // 2:])
        wp, bp = G.pf.pack_geglu_weight(w, b)
        old = G.pf.gemm(n16, wp, bp, mode="geglu")
    else:
        S = M // 4
        y = G.pf.gemm_ln(x.view(4, S, C), st, *f, mode="trans", tokens=S)
        ref = proj.view(4, S, N).transpose(1, 2)
        old = G.pf.gemm(n16.view(4, S, C), w, b, mode="trans", tokens=S)
    scale = ref.abs().max().item()
    e_new, e_old = (y.float() - ref).abs().max().item() / scale, (old.float() - ref).abs().max().item() / scale
    assert y.shape == old.shape and e_new <= 3e-3, (e_new, e_old)
    assert e_new <= 1.5 * e_old + 5e-4, (e_new, e_old)          # the folded form skips the fp16 rounding of LayerNorm(x): it is not less accurate


@pytest.mark.parametrize("sigmas", [50.0, 100.0])
@pytest.mark.parametrize("M,C", [(1024, 320), (1024, 1280)])
def test_row_statistics_with_a_large_common_offset(G, sigmas, M, C):
    """The row records are one-pass (sum, sum of squares) in fp32: var = E[x^2] - mean^2 cancels when |mean| >> std.  Rows whose mean is 50 / 100 standard
    deviations away from zero (far beyond what a residual stream holds) still give rstd to 1e-3 / 4e-3 relative -- the fp16 output rounding is 5e-4 --
    and the folded GEMM stays within the fp16 bound of the fp32 reference; the validity range (|mean| / std up to ~100) is stated in DESIGN.md."""
    g = torch.Generator().manual_seed(int(sigmas) + C)
    a = torch.randn(M, 320, generator=g).half().cuda()
    w = (torch.randn(C, 320, generator=g) * 320 ** -0.5 * 0.05).half().cuda()          # output std ~0.05
    b = torch.full((C,), 0.05 * sigmas).half().cuda()                                   # + a common offset of `sigmas` standard deviations
    x = G.pf.gemm(a, w, b, rowstats=True)
    st = G.pf.ln_stat(x, 1e-5)
    assert st is not None
    xf = x.double()
    mean, var = xf.mean(1), xf.var(1, unbiased=False)
    assert float((mean.abs() / var.sqrt()).min()) > 0.8 * sigmas
    rstd = (var + 1e-5).rsqrt()
    rel = ((st[:, 0].double() - rstd).abs() / rstd).max().item()
    assert rel <= (1e-3 if sigmas <= 50 else 4e-3), rel
    gamma, beta = torch.ones(C).half().cuda(), torch.zeros(C).half().cuda()
    w2 = (torch.randn(640, C, generator=g) * C ** -0.5).half().cuda()
    y = G.pf.gemm_ln(x, st, *G.pf.fold_ln_weights(w2, None, gamma, beta))
    ref = F.layer_norm(x.float(), (C,), None, None, 1e-5) @ w2.float().T
    assert (y.float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()


def test_transformer_block_folded_vs_unfolded_vs_fp32(G):
    U = G.unet
    torch.manual_seed(0)
    blk = U.BasicTransformerBlock(640, 1024, 10, 64)
    U.synthetic_init_(blk, 3)
    with torch.no_grad():
        for n_ in (blk.norm1, blk.norm2, blk.norm3):
            n_.weight.add_(0.2 * torch.randn_like(n_.weight))
            n_.bias.add_(0.1 * torch.randn_like(n_.bias))
    blk = blk.cuda().half().eval()
    B, S = 16, 1024
    x = _producer(G, B * S, 320, 640, 1).view(B, S, 640)
    x._gsw_rowstats = None
    xr = _producer(G, B * S, 320, 640, 1).view(B, S, 640)           # the same values, with the row records
    ctx = torch.randn(B, 77, 1024, device="cuda").half()
    prev = G.pf.FOLD_LN
    with torch.no_grad():
        y_f = blk(xr, ctx)
        G.pf.FOLD_LN = False
        try:
            y_u = blk(xr, ctx)
        finally:
            G.pf.FOLD_LN = prev
        import copy
        bf = copy.deepcopy(blk).float()
        U.FUSED_KERNELS = False
        try:
            ref = bf(xr.float(), ctx.float())
        finally:
            U.FUSED_KERNELS = True
    scale = ref.abs().max().item()
    assert (y_f.float() - ref).abs().max().item() <= 6e-3 * scale and (y_u.float() - ref).abs().max().item() <= 6e-3 * scale
    assert (y_f.float() - y_u.float()).abs().max().item() <= 6e-3 * scale
