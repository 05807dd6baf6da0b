"""GPU: the small-batch kernels (csrc/gswm_small.hip) against fp32 torch references -- one-launch GroupNorm, the time-embedding table, the input
packing kernel, the direct conv_out -- and the forward that uses them against the forward that does not."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, unet, _native, codec
    old = pf.GN_FUSED_MAX_PIXELS
    pf.GN_FUSED_MAX_PIXELS = 1 << 20          # the kernel is tested on every lattice it accepts, not only where the dispatch prefers it
    yield types.SimpleNamespace(pf=pf, unet=unet, N=_native, codec=codec)
    pf.GN_FUSED_MAX_PIXELS = old


def _gn_ref(x, x2, gamma, beta, act):
    xx = x.float() if x2 is None else torch.cat([x.float(), x2.float()], dim=1)
    ref = F.group_norm(xx, 32, gamma.float(), beta.float(), 1e-5)
    return F.silu(ref) if act else ref


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,C,C2,H,W", [(1, 320, 0, 64, 64), (2, 640, 0, 32, 32), (1, 1280, 0, 8, 8), (2, 1280, 0, 16, 16), (1, 320, 320, 32, 32),
                                         (2, 1280, 640, 16, 16), (1, 1280, 1280, 8, 8), (1, 640, 640, 32, 32), (3, 64, 0, 4, 6), (1, 128, 64, 5, 7), (1, 320, 0, 48, 96),
                                         (1, 1280, 640, 32, 32)])
@pytest.mark.parametrize("act", [True, False])
def test_groupnorm_fused_vs_torch_fp32(G, dtype, B, C, C2, H, W, act):
    """one workgroup per (image, group): single source, channel concatenations whose groups straddle the two sources (1280 + 640 -> 60-channel groups),
    32 rows per row lane (the largest variant), odd lattices, both output forms"""
    g = torch.Generator().manual_seed(C + C2 + H)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to(dtype).cuda()
    x2 = (torch.randn(B, C2, H, W, generator=g) * 0.7 - 0.5).to(dtype).cuda() if C2 else None
    Ct = C + C2
    gamma = (1 + 0.2 * torch.randn(Ct, generator=g)).to(dtype).cuda()
    beta = (0.2 * torch.randn(Ct, generator=g)).to(dtype).cuda()
    assert G.pf._gn_fused_ok(B, H, W, Ct, 32), "shape list is meant for the fused kernel"
    ref = _gn_ref(x, x2, gamma, beta, act)
    xp = G.pf.PF.from_nchw(x)
    xp2 = G.pf.PF.from_nchw(x2) if C2 else None
    y = G.pf.groupnorm_pf2(xp, xp2, gamma, beta, 32, 1e-5, act=act)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (y.to_nchw().float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    grid = y.grid.float()
    assert grid[:, 0].abs().max() == 0 and grid[:, -1].abs().max() == 0 and grid[:, :, 0].abs().max() == 0 and grid[:, :, -1].abs().max() == 0
    if not C2:
        t = G.pf.groupnorm_pf(xp, gamma, beta, 32, 1e-5, act=act, tokens=True)
        assert t.shape == (B, H * W, Ct)
        assert torch.equal(t.view(B, H, W, Ct), y.interior)
    # the two-launch form of the same GroupNorm agrees to a rounding of the output
    old = G.pf.GN_FUSED_MAX_WGS
    G.pf.GN_FUSED_MAX_WGS = 0
    try:
        y_old = G.pf.groupnorm_pf2(xp, xp2, gamma, beta, 32, 1e-5, act=act)
    finally:
        G.pf.GN_FUSED_MAX_WGS = old
    assert (y.to_nchw().float() - y_old.to_nchw().float()).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_groupnorm_fused_large_offset(G):
    """rows offset by 50 sigma: the fused kernel's variance is a true two-pass (mean first, then centred squares)"""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(1, 320, 32, 32, generator=g) * 0.05 + 2.5).half().cuda()
    gamma = torch.ones(320).half().cuda()
    beta = torch.zeros(320).half().cuda()
    ref = F.group_norm(x.float(), 32, None, None, 1e-5)
    y = G.pf.groupnorm_pf(G.pf.PF.from_nchw(x), gamma, beta, 32, 1e-5, act=False)
    assert (y.to_nchw().float() - ref).abs().max().item() <= 6e-3 * ref.abs().max().item()


def test_groupnorm_fused_validation(G):
    lib = G.N.lib()
    x = torch.zeros(4096, dtype=torch.float16, device="cuda")
    p = x.data_ptr()
    assert lib.gsw_groupnorm_pf_fused(None, None, 0, p, p, p, 1, 8, 8, 64, 32, 1e-5, 1, 0, G.N.GSW_F16, None) == G.N.GSW_ERR_BAD_ARG
    assert lib.gsw_groupnorm_pf_fused(p, None, 0, p, p, p, 1, 8, 8, 64, 32, 1e-5, 1, 0, G.N.GSW_F32, None) == G.N.GSW_ERR_BAD_ARG
    assert lib.gsw_groupnorm_pf_fused(p, None, 0, p, p, p, 1, 8, 8, 32, 32, 1e-5, 1, 0, G.N.GSW_F16, None) == G.N.GSW_ERR_UNSUPPORTED      # one channel per group
    assert lib.gsw_groupnorm_pf_fused(p, None, 0, p, p, p, 1, 8, 512, 320, 32, 1e-5, 1, 0, G.N.GSW_F16, None) == G.N.GSW_ERR_UNSUPPORTED    # a row does not fit
    assert lib.gsw_groupnorm_pf_fused(p, p, 12, p, p, p, 1, 8, 8, 64, 32, 1e-5, 1, 0, G.N.GSW_F16, None) == G.N.GSW_ERR_BAD_ARG             # Ca % 8
    assert not G.pf._gn_fused_ok(64, 64, 64, 320, 32) and not G.pf._gn_fused_ok(1, 512, 512, 128, 32) and not G.pf._gn_fused_ok(1, 64, 64, 640, 32)


def test_gather_rows(G):
    lib = G.N.lib()
    table = torch.randn(1000, 20160, device="cuda").half()
    for idx in (torch.tensor([3, 999, 0, 500], device="cuda"), torch.tensor([1500, -4], device="cuda")):
        out = torch.empty((idx.numel(), 20160), dtype=torch.float16, device="cuda")
        G.N.check(lib.gsw_gather_rows(table.data_ptr(), 20160 * 2, 1000, idx.data_ptr(), 1, out.data_ptr(), 20160 * 2, idx.numel(), 20160 * 2, None))
        assert torch.equal(out, table[idx.clamp(0, 999)])
    one = torch.full((), 77, device="cuda", dtype=torch.int64)
    out = torch.empty((3, 20160), dtype=torch.float16, device="cuda")
    G.N.check(lib.gsw_gather_rows(table.data_ptr(), 20160 * 2, 1000, one.data_ptr(), 0, out.data_ptr(), 20160 * 2, 3, 20160 * 2, None))
    assert torch.equal(out, table[77].expand(3, -1))
    assert lib.gsw_gather_rows(table.data_ptr(), 20160 * 2, 1000, one.data_ptr(), 0, out.data_ptr(), 20160 * 2, 3, 20160 * 2 + 2, None) == G.N.GSW_ERR_BAD_ARG


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Cin,H,W", [(1, 4, 64, 64), (3, 4, 8, 12), (2, 9, 5, 7)])
def test_nchw_to_pf(G, dtype, B, Cin, H, W):
    x = torch.randn(B, Cin, H, W, device="cuda").to(dtype)
    y = G.pf.PF.empty(B, H, W, 64, dtype, x.device)
    y.buf.fill_(7.0)
    G.N.check(G.N.lib().gsw_nchw_to_pf(x.data_ptr(), y.rows.data_ptr(), B, Cin, H, W, 64, G.codec._dt(dtype), None))
    assert torch.equal(y.interior[..., :Cin], x.permute(0, 2, 3, 1))
    assert y.interior[..., Cin:].abs().max() == 0
    g = y.grid
    assert g[:, 0].abs().max() == 0 and g[:, -1].abs().max() == 0 and g[:, :, 0].abs().max() == 0 and g[:, :, -1].abs().max() == 0
    assert (y.buf[: y.G] == 7.0).all() and (y.buf[y.G + y.M:] == 7.0).all()          # guard rows are not this kernel's to write


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,C,N,H,W", [(1, 320, 4, 64, 64), (2, 320, 4, 8, 8), (3, 64, 3, 5, 7), (1, 128, 16, 6, 6), (2, 96, 8, 9, 4)])
def test_conv3x3_pf_nchw_vs_torch_fp32(G, dtype, B, C, N, H, W):
    g = torch.Generator().manual_seed(C + N + H)
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * (1.0 / (C * 9)) ** 0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    y = torch.empty((B, N, H, W), dtype=dtype, device="cuda")
    xp = G.pf.PF.from_nchw(x)
    G.N.check(G.N.lib().gsw_conv3x3_pf_nchw(xp.rows.data_ptr(), G.pf.pack_conv_weight(w).data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, C, N, G.codec._dt(dtype), None))
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert (y.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert G.N.lib().gsw_conv3x3_pf_nchw(xp.rows.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, H, W, C, 17, G.codec._dt(dtype), None) == G.N.GSW_ERR_UNSUPPORTED


def test_unet_forward_with_and_without_the_small_batch_kernels(G):
    """One- and two-row forwards through the time-embedding table / fused GroupNorm / direct conv_out against the same forward with those switched off
    (the chain of GEMMs, the two-launch GroupNorm, the 64-column conv_out, the engine for the 8 x 8 level's dense linears): the same function up to roundings
    of intermediate tensors."""
    U = G.unet
    m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
    g = torch.Generator().manual_seed(3)
    U.FALLBACKS.clear()
    for B in (1, 2):
        x = torch.randn(B, 4, 64, 64, generator=g).half().cuda()
        c = torch.randn(B, 77, 1024, generator=g).half().cuda()
        t = torch.full((), 481, device="cuda", dtype=torch.int64)
        G.pf.GN_FUSED_MAX_PIXELS = 1024                # the dispatch's own rule
        with torch.no_grad():
            y_new = m(x, t, c)
            old = (U.TEMB_TABLE, G.pf.GN_FUSED_MAX_WGS, U.CONV_OUT_DIRECT_MAX_PIXELS, G.pf.SMALL_GEMM_MAX_ROWS)
            U.TEMB_TABLE, G.pf.GN_FUSED_MAX_WGS, U.CONV_OUT_DIRECT_MAX_PIXELS, G.pf.SMALL_GEMM_MAX_ROWS = False, 0, 0, 0
            try:
                y_old = m(x, t, c)
                y_float_t = m(x, t.float(), c)              # a float timestep takes the chain as well
            finally:
                U.TEMB_TABLE, G.pf.GN_FUSED_MAX_WGS, U.CONV_OUT_DIRECT_MAX_PIXELS, G.pf.SMALL_GEMM_MAX_ROWS = old
        scale = y_old.float().abs().max().item()
        assert (y_new.float() - y_old.float()).abs().max().item() <= 1e-2 * scale
        assert torch.equal(y_old, y_float_t)
        assert not U.FALLBACKS
        # per-image timesteps gather per-image rows
        if B == 2:
            tt = torch.tensor([481, 21], device="cuda")
            with torch.no_grad():
                y2 = m(x, tt, c)
                y2b = m(x[1:], tt[1:], c[1:])
            assert (y2[0].float() - y_new[0].float()).abs().max().item() <= 1e-2 * scale
            assert (y2[1].float() - y2b[0].float()).abs().max().item() <= 1e-2 * scale


# ---------------------------------------------------------------------------------------------------------------------------------------------
# gsw_gemm_small: the small-M dense kernel (K split four ways inside the workgroup), every configuration and epilogue against fp32
# ---------------------------------------------------------------------------------------------------------------------------------------------
_MODES = {"plain": 0, "geglu": 1, "trans": 2, "tok2pf": 3}


def _small(G, x, w, bias, y, mode, cfg, *, resid=None, S=0, Wimg=0, ln=None, rowstats=None, ldr=None, ldy=None):
    import ctypes as C
    M, K = x.shape
    Nn = w.shape[0]
    ncols = Nn // 2 if mode == "geglu" else Nn
    slots = C.c_int(0)
    rec, nslots, eps, u, v = ln if ln is not None else (None, 0, 0.0, None, None)
    rc = G.N.lib().gsw_gemm_small(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr() if bias is not None else None,
                                  resid.data_ptr() if resid is not None else None, ldr if ldr is not None else ncols, y.data_ptr(), ldy if ldy is not None else ncols,
                                  M, K, Nn, _MODES[mode], S, Wimg, rec.data_ptr() if rec is not None else None, nslots, eps,
                                  u.data_ptr() if u is not None else None, v.data_ptr() if v is not None else None,
                                  rowstats.data_ptr() if rowstats is not None else None, rowstats.numel() if rowstats is not None else 0, C.byref(slots), cfg,
                                  G.codec._dt(x.dtype), None)
    return rc, slots.value


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("M,K,N", [(64, 1280, 1280), (256, 320, 640), (80, 128, 48), (1024, 640, 320), (16, 5120, 160)])
def test_gemm_small_plain_trans_vs_fp32(G, dtype, cfg, M, K, N):
    g = torch.Generator().manual_seed(M + K + N + cfg)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    ref = x.float() @ w.float().t() + b.float()
    y = torch.empty(M, N, dtype=dtype, device="cuda")
    rc, _ = _small(G, x, w, b, y, "plain", cfg)
    assert rc == 0
    assert (y.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    rc, _ = _small(G, x, w, None, y, "plain", cfg, resid=r)
    assert rc == 0
    ref2 = (x.float() @ w.float().t()).to(dtype).float() + r.float()
    assert (y.float() - ref2).abs().max().item() <= tol * ref2.abs().max().item()
    # two runs are bit-identical (the four partial tiles are added in wave order)
    y2 = torch.empty_like(y)
    _small(G, x, w, None, y2, "plain", cfg, resid=r)
    assert torch.equal(y, y2)
    # transposed output, two "images" of M / 2 tokens when that is a multiple of 4
    S = M // 2 if (M // 2) % 4 == 0 else M
    yt = torch.empty(M // S, N, S, dtype=dtype, device="cuda")
    rc, _ = _small(G, x, w, b, yt, "trans", cfg, S=S)
    assert rc == 0
    assert (yt.float() - ref.view(M // S, S, N).transpose(1, 2)).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
@pytest.mark.parametrize("M,K,N", [(64, 1280, 2560), (256, 320, 320), (48, 128, 96)])
def test_gemm_small_geglu_vs_fp32(G, dtype, cfg, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    I = N // 2
    if I % 8:
        pytest.skip("packing needs 8-output blocks")
    wp, bp = G.pf.pack_geglu_weight(w, b)
    y = torch.empty(M, I, dtype=dtype, device="cuda")
    rc, _ = _small(G, x, wp, bp, y, "geglu", cfg)
    assert rc == 0
    pre = x.float() @ w.float().t() + b.float()
    ref = pre[:, :I].to(dtype).float() * F.gelu(pre[:, I:].to(dtype).float()).to(dtype).float()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("cfg", [0, 1, 2, 3])
def test_gemm_small_tok2pf_in_place(G, cfg):
    """proj_out: tokens -> interior rows of the PF tensor + residual, written in place"""
    B, H, W, C = 2, 8, 16, 320
    g = torch.Generator().manual_seed(cfg)
    tok = torch.randn(B * H * W, C, generator=g).half().cuda()
    w = (torch.randn(C, C, generator=g) * C ** -0.5).half().cuda()
    b = torch.randn(C, generator=g).half().cuda()
    x = G.pf.PF.from_nchw(torch.randn(B, C, H, W, generator=g).half().cuda())
    before = x.interior.float().clone()
    rc, _ = _small(G, tok, w, b, x.rows, "tok2pf", cfg, resid=x.rows, S=H * W, Wimg=W, ldr=C, ldy=C)
    assert rc == 0
    ref = (tok.float() @ w.float().t() + b.float()).half().float().view(B, H, W, C) + before
    assert (x.interior.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    gr = x.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg_p,cfg_c", [(0, 0), (1, 3), (3, 1), (2, 2)])
@pytest.mark.parametrize("M,C,N", [(64, 1280, 1280), (256, 320, 640), (1024, 640, 2560)])
def test_gemm_small_ln_fold_from_raw_records(G, dtype, cfg_p, cfg_c, M, C, N):
    """producer (plain + residual) leaves row records; the consumer folds LayerNorm from them (plain / transposed / GEGLU): against fp32 LN -> linear of
    the producer's stored output, rows with a 3 sigma offset"""
    g = torch.Generator().manual_seed(M + C + N)
    x0 = torch.randn(M, C, generator=g).to(dtype).cuda()
    wp_ = (torch.randn(C, C, generator=g) * C ** -0.5).to(dtype).cuda()
    res = (torch.randn(M, C, generator=g) + 3.0 * torch.randn(M, 1, generator=g)).to(dtype).cuda()
    xs = torch.empty(M, C, dtype=dtype, device="cuda")
    rec = torch.empty(M * ((C + 31) // 32) * 2, dtype=torch.float32, device="cuda")
    rc, slots = _small(G, x0, wp_, None, xs, "plain", cfg_p, resid=res, rowstats=rec)
    assert rc == 0 and slots == (C + (32 if cfg_p == 3 else 64) - 1) // (32 if cfg_p == 3 else 64)
    r = rec[: M * slots * 2].view(M, slots, 2).double().sum(dim=1)
    assert torch.allclose(r[:, 0], xs.double().sum(dim=1), rtol=1e-4, atol=1e-2)
    assert torch.allclose(r[:, 1], (xs.double() ** 2).sum(dim=1), rtol=1e-4, atol=1e-2)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    beta = (0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    w = (torch.randn(N, C, generator=g) * C ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    ln = F.layer_norm(xs.float(), (C,), gamma.float(), beta.float(), 1e-5)
    ref = ln @ w.float().t() + b.float()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    wf, u, v = G.pf.fold_ln_weights(w, b, gamma, beta)
    y = torch.empty(M, N, dtype=dtype, device="cuda")
    rc, _ = _small(G, xs, wf, None, y, "plain", cfg_c, ln=(rec, slots, 1e-5, u, v))
    assert rc == 0
    assert (y.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    yt = torch.empty(1, N, M, dtype=dtype, device="cuda")
    rc, _ = _small(G, xs, wf, None, yt, "trans", cfg_c, S=M, ln=(rec, slots, 1e-5, u, v))
    assert rc == 0
    assert (yt[0].float() - ref.t()).abs().max().item() <= tol * ref.abs().max().item()
    wg, ug, vg = G.pf.fold_ln_weights(w, b, gamma, beta, geglu=True)
    yg = torch.empty(M, N // 2, dtype=dtype, device="cuda")
    rc, _ = _small(G, xs, wg, None, yg, "geglu", cfg_c, ln=(rec, slots, 1e-5, ug, vg))
    assert rc == 0
    refg = ref[:, : N // 2].to(dtype).float() * F.gelu(ref[:, N // 2:].to(dtype).float()).to(dtype).float()
    assert (yg.float() - refg).abs().max().item() <= 2 * tol * max(1.0, refg.abs().max().item())


def test_gemm_small_validation(G):
    lib = G.N.lib()
    assert lib.gsw_gemm_small_config(64, 1280, 1280, 0) == 3 and lib.gsw_gemm_small_config(4096, 320, 320, 0) == 0 and lib.gsw_gemm_small_config(256, 1280, 1280, 0) == 2
    assert lib.gsw_gemm_small_config(100, 1280, 1280, 0) == -1 and lib.gsw_gemm_small_config(64, 96, 1280, 0) == -1 and lib.gsw_gemm_small_config(64, 1280, 1288, 0) == -1
    assert lib.gsw_gemm_small_config(32768, 320, 320, 0) == -1
    x = torch.zeros(64, 1280, dtype=torch.float16, device="cuda")
    y = torch.zeros(64, 1280, dtype=torch.float16, device="cuda")
    assert _small(G, x, x, None, y, "plain", 7)[0] == G.N.GSW_ERR_UNSUPPORTED
    assert _small(G, x[:, 8:], x, None, y, "plain", 0)[0] != 0            # K = 1272
    assert _small(G, x, x, None, y, "trans", 0, S=0)[0] == G.N.GSW_ERR_UNSUPPORTED
    assert _small(G, x, x, None, y, "geglu", 0, resid=y)[0] == G.N.GSW_ERR_UNSUPPORTED
    u = torch.zeros(1280, device="cuda")
    assert _small(G, x, x, y[0], y, "plain", 0, ln=(u, 1, 1e-5, u, u))[0] == G.N.GSW_ERR_BAD_ARG      # bias and LayerNorm fold exclude each other


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,Sq,Sk,D", [(1, 5, 4096, 4096, 64), (1, 10, 1024, 1024, 64), (2, 3, 128, 2112, 64), (1, 2, 256, 2752, 64), (1, 8, 1024, 2304, 40), (2, 5, 2048, 512, 64)])
def test_attention_key_split_vs_unsplit_and_fp32(G, dtype, B, H, Sq, Sk, D):
    """gsw_attention_ws: few query tiles against many key tiles (one image's self-attention) run as several workgroups per query tile over disjoint key ranges +
    gsw_attn_combine_kernel, from 32 key tiles up.  Key-tile counts that do not divide by the split count (64 -> 21, 21, 22; 33, 43 and 36 tiles over 8 workgroups),
    head_dim 40, both dtypes, two shapes below the threshold (same bits as unsplit); against the unsplit kernel (pf.ATTN_KEY_SPLIT = False) and fp32 torch; deterministic."""
    pf = G.pf
    g = torch.Generator().manual_seed(Sq + Sk + H)
    q = (torch.randn(B, Sq, H * D, generator=g) * 2.0).to(dtype).cuda()
    k = torch.randn(B, Sk, H * D, generator=g).to(dtype).cuda()
    v = torch.randn(B, Sk, H * D, generator=g).to(dtype).cuda()
    vt = v.transpose(1, 2).contiguous()
    assert pf.ATTN_KEY_SPLIT
    a = pf.attention(q, k, vt, H)
    a2 = pf.attention(q, k, vt, H)
    pf.ATTN_KEY_SPLIT = False
    try:
        b = pf.attention(q, k, vt, H)
    finally:
        pf.ATTN_KEY_SPLIT = True
    qf, kf, vf = (t.float().view(B, t.shape[1], H, D).transpose(1, 2) for t in (q, k, v))
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * D ** -0.5, dim=-1) @ vf).transpose(1, 2).reshape(B, Sq, H * D)
    tol = (4e-3 if dtype == torch.float16 else 2e-2) * max(1.0, ref.abs().max().item())
    assert torch.equal(a, a2)
    assert (a.float() - ref).abs().max().item() <= tol and (b.float() - ref).abs().max().item() <= tol
    assert (a.float() - b.float()).abs().max().item() <= tol
    if Sk >= 2048:
        assert not torch.equal(a, b) or dtype == torch.bfloat16          # the split path really ran (another summation order; bf16 may round both to the same bits)
    else:
        assert torch.equal(a, b)


def test_attention_ws_validation_and_small_workspace_falls_back(G):
    lib = G.N.lib()
    q = torch.randn(1, 1024, 640, device="cuda").half()
    vt = q.transpose(1, 2).contiguous()
    out = torch.empty_like(q)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    args = (q.data_ptr(), q.data_ptr(), vt.data_ptr(), out.data_ptr(), 1, 10, 64, 1024, 1024, 1024, 640, 640, 640, 0.125, G.N.GSW_F16)
    assert lib.gsw_attention_ws(*args, None, 16, None) == G.N.GSW_ERR_BAD_ARG and lib.gsw_attention_ws(*args, ws.data_ptr() + 8, 1024, None) == G.N.GSW_ERR_BAD_ARG
    assert lib.gsw_attention_ws(*args, ws.data_ptr(), -1, None) == G.N.GSW_ERR_BAD_ARG
    # a shape below the key-tile threshold (and a buffer too small for any split of it): the unsplit kernel -- same bits as no buffer at all
    assert lib.gsw_attention_ws(*args, ws.data_ptr(), ws.numel(), None) == 0
    torch.cuda.synchronize()
    small = out.clone()
    assert lib.gsw_attention_ws(*args, None, 0, None) == 0
    torch.cuda.synchronize()
    assert torch.equal(small, out)
