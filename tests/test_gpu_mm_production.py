"""GPU: the matmul engine AT ITS PRODUCTION TILING against fp32 torch -- the launches that carry the end-to-end bench (batch 64):
256-row tiles (MT = 4), >= 256 output tiles, several tiles per persistent workgroup, the XCD panel order, every epilogue, the real
convolution shapes of the SD 2.1 UNet (`pipe(...)` at the reference's extract.py:66-69), each also on the forced 128-row tiling
(gsw_mm_config).  Then whole models: the full SD 2.1-shaped UNet (865.9 M parameters) and the SD VAE against the fp32 torch forward of
the same module, with an ABSOLUTE bound at the output's scale -- the lossless round trip cannot see an eps-model error (sampling and
inversion share the model), these tests can."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, unet, vae, _native
    old = pf.SMALL_GEMM_MAX_ROWS
    pf.SMALL_GEMM_MAX_ROWS = 0          # this module tests the matmul ENGINE (at <= 128 rows pf.gemm would otherwise take gsw_gemm_small: tests/test_gpu_small.py)
    yield types.SimpleNamespace(pf=pf, unet=unet, vae=vae, lib=_native.lib())
    pf.SMALL_GEMM_MAX_ROWS = old


@pytest.fixture(params=[256, 128, 512], ids=["BM256", "BM128", "WIDE256x320"])      # 512: the 256 x 320 tile wherever it is legal (N % 320 == 0, dense / PF / GEGLU epilogues)
def tile_rows(request, G):
    assert G.lib.gsw_mm_config(request.param, -1) == 0
    yield request.param
    assert G.lib.gsw_mm_config(0, -1) == 0


def _tiles(M, N, bm):
    return ((M + bm - 1) // bm) * ((N + 159) // 160)


TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}          # one rounding of the storage dtype at the output magnitude


def _rel(y, ref):
    return (y.float() - ref).abs().max().item() / ref.abs().max().item()


# (B, C_in, C_out, H, W, ksize, stride): the seven conv shapes of profiles/r02h_conv_engine_vs_halo.txt at batches that give the 256-row
# tiling >= 256 tiles and an uneven number of tiles per workgroup
CONV_SHAPES = [(18, 320, 320, 64, 64, 3, 1), (9, 960, 320, 64, 64, 3, 1), (40, 640, 640, 32, 32, 3, 1), (24, 1920, 640, 32, 32, 3, 1),
               (72, 1280, 1280, 16, 16, 3, 1), (40, 2560, 1280, 16, 16, 3, 1), (136, 1280, 1280, 8, 8, 3, 1),
               (65, 640, 640, 32, 32, 3, 2), (33, 320, 320, 64, 64, 3, 2), (25, 640, 1280, 32, 32, 1, 1)]


@pytest.mark.parametrize("B,C,N,H,W,k,stride", CONV_SHAPES)
def test_conv_production_tiling_vs_fp32(G, tile_rows, B, C, N, H, W, k, stride):
    dtype = torch.float16
    Ho, Wo = H // stride if stride == 2 else H, W // stride if stride == 2 else W
    Hi, Wi = (H, W) if stride == 1 else (H, W)
    if tile_rows == 256:
        assert _tiles(B * Ho * Wo, N, 256) >= 256 and _tiles(B * Ho * Wo, N, 256) % 256 != 0
    g = torch.Generator().manual_seed(C + N + H + k)
    x = torch.randn(B, C, Hi, Wi, generator=g).to(dtype).cuda()
    w = (torch.randn(N, C, k, k, generator=g) * (1.0 / (C * k * k)) ** 0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    rb = torch.randn(B, N, generator=g).to(dtype).cuda()
    res = torch.randn(B, N, Ho, Wo, generator=g).to(dtype).cuda()
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=k // 2, stride=stride) + rb.float()[:, :, None, None] + res.float()
    y = G.pf.conv_pf(G.pf.PF.from_nchw(x), G.pf.pack_conv_weight(w), b, ksize=k, stride=stride, rowbias=rb, resid=G.pf.PF.from_nchw(res))
    assert (y.B, y.H, y.W, y.C) == (B, Ho, Wo, N)
    assert _rel(y.to_nchw(), ref) <= TOL[dtype]
    gr = y.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("B,C,N,C1,C2,H,W", [(40, 1280, 1280, 1280, 1280, 16, 16), (20, 640, 640, 1280, 640, 32, 32), (10, 320, 320, 320, 320, 64, 64)])
def test_resnet_three_segment_launch_vs_fp32(G, tile_rows, B, C, N, C1, C2, H, W):
    """conv2 + conv_shortcut over cat(x, skip) + bias in ONE launch (three K segments), up-block shapes"""
    dtype = torch.float16
    if tile_rows == 256:
        assert _tiles(B * H * W, N, 256) >= 256
    g = torch.Generator().manual_seed(C + N + C1 + C2)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(B, C, H, W).to(dtype).cuda()
    x1, x2 = rnd(B, C1, H, W).to(dtype).cuda(), rnd(B, C2, H, W).to(dtype).cuda()
    w3 = (rnd(N, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).to(dtype).cuda()
    w1 = (rnd(N, C1 + C2) * (1.0 / (C1 + C2)) ** 0.5).to(dtype).cuda()
    b = rnd(N).to(dtype).cuda()
    ref = F.conv2d(x.float(), w3.float(), b.float(), padding=1) + F.conv2d(torch.cat([x1, x2], 1).float(), w1.float()[:, :, None, None])
    P = G.pf.PF.from_nchw
    y = G.pf.conv3x3_res_pf(P(x), torch.cat([G.pf.pack_conv_weight(w3), w1], dim=1).contiguous(), b, x1=P(x1), x2=P(x2))
    assert _rel(y.to_nchw(), ref) <= TOL[dtype]
    assert y.grid[:, 0].abs().max() == 0 and y.grid[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("B,H,W,C,N", [(72, 16, 16, 1280, 1280), (40, 32, 32, 640, 640)])
def test_up2x_production_tiling_vs_fp32(G, tile_rows, B, H, W, C, N):
    dtype = torch.float16
    if tile_rows == 256:
        assert _tiles(B * H * W, N, 256) >= 256
    g = torch.Generator().manual_seed(B * 100 + H)
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dtype).cuda()
    b = (0.1 * torch.randn(N, generator=g)).to(dtype).cuda()
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    y = G.pf.conv_up2x_pf(G.pf.PF.from_nchw(x), G.pf.pack_upsample_weight(w), b)
    assert _rel(y.to_nchw(), ref) <= 4e-3          # the pre-summed weights are rounded once more to fp16
    gr = y.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(64 * 4096 + 40, 320, 320), (36 * 1024, 640, 1920), (20 * 4096, 320, 640), (66 * 256, 1280, 1280)])
def test_dense_rows_production_tiling_vs_fp32(G, tile_rows, dtype, M, K, N):
    if tile_rows == 256:
        assert _tiles(M, N, 256) >= 256
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    ref = x.float() @ w.float().T + b.float()
    assert _rel(G.pf.gemm(x, w, b), ref) <= TOL[dtype]
    assert _rel(G.pf.gemm(x, w, b, resid=r), ref + r.float()) <= 2 * TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,I", [(10 * 4096 + 8, 320, 1280), (20 * 1024, 640, 2560), (40 * 256, 1280, 5120)])
def test_geglu_production_tiling_vs_fp32(G, tile_rows, dtype, M, K, I):
    if tile_rows == 256:
        assert _tiles(M, 2 * I, 256) >= 256
    g = torch.Generator().manual_seed(M + I)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = (0.5 * torch.randn(2 * I, generator=g)).to(dtype).cuda()
    wp, bp = G.pf.pack_geglu_weight(w, b)
    y = G.pf.gemm(x, wp, bp, mode="geglu")
    h = (x.float() @ w.float().T + b.float()).to(dtype).float()          # torch materialises the projection in the storage dtype
    ref = h[:, :I] * F.gelu(h[:, I:])
    assert (y.float() - ref).abs().max().item() <= 2 * TOL[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,S,K,N", [(64, 4096, 320, 320), (66, 1024, 640, 640), (34, 256, 1280, 1280)])
def test_transposed_projection_production_tiling_vs_fp32(G, tile_rows, dtype, B, S, K, N):
    """EPI 3 (the attention kernel's V^T operand) on the 12-wave variant"""
    if tile_rows == 256:
        assert _tiles(B * S, N, 256) >= 256
    g = torch.Generator().manual_seed(B + S + N)
    x = torch.randn(B, S, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    y = G.pf.gemm(x, w, b, mode="trans", tokens=S)
    ref = (x.float() @ w.float().T + b.float()).transpose(1, 2)
    assert y.shape == (B, N, S) and _rel(y, ref) <= TOL[dtype]


@pytest.mark.parametrize("B,H,W,K,N", [(32, 64, 64, 320, 320), (66, 32, 32, 640, 640)])
def test_tok2pf_production_tiling_vs_fp32(G, tile_rows, B, H, W, K, N):
    dtype = torch.float16
    if tile_rows == 256:
        assert _tiles(B * H * W, N, 256) >= 256
    g = torch.Generator().manual_seed(B + H + N)
    tok = torch.randn(B, H * W, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    base = torch.randn(B, N, H, W, generator=g).to(dtype).cuda()
    x = G.pf.PF.from_nchw(base)
    G.pf.gemm(tok, w, b, resid=x.rows, mode="tok2pf", tokens=H * W, width=W, out=x.rows)
    ref = (tok.float() @ w.float().T + b.float()).view(B, H, W, N).permute(0, 3, 1, 2) + base.float()
    assert _rel(x.to_nchw(), ref) <= 2 * TOL[dtype]
    gr = x.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("split_mask", [0, 15])
def test_every_epilogue_on_both_wave_layouts(G, split_mask):
    """8-wave (split_mask 0) and 12-wave (15) variants of every epilogue on one mid-size launch each"""
    assert G.lib.gsw_mm_config(-1, split_mask) == 0
    try:
        dtype = torch.float16
        g = torch.Generator().manual_seed(split_mask)
        M, K, N = 40 * 1024 + 24, 640, 640
        x = torch.randn(M, K, generator=g).to(dtype).cuda()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
        b = torch.randn(N, generator=g).to(dtype).cuda()
        r = torch.randn(M, N, generator=g).to(dtype).cuda()
        ref = x.float() @ w.float().T + b.float()
        assert _rel(G.pf.gemm(x, w, b, resid=r), ref + r.float()) <= 2 * TOL[dtype]
        wp, bp = G.pf.pack_geglu_weight(w, b)
        h = ref.to(dtype).float()
        refg = h[:, : N // 2] * F.gelu(h[:, N // 2:])
        assert (G.pf.gemm(x, wp, bp, mode="geglu").float() - refg).abs().max().item() <= 2 * TOL[dtype] * max(1.0, refg.abs().max().item())
        xs = x[: 40 * 1024].view(40, 1024, K)
        assert _rel(G.pf.gemm(xs, w, b, mode="trans", tokens=1024), (ref[: 40 * 1024].view(40, 1024, N)).transpose(1, 2)) <= TOL[dtype]
        xc = torch.randn(40, 640, 32, 32, generator=g).to(dtype).cuda()
        wc = (torch.randn(640, 640, 3, 3, generator=g) * (1.0 / (9 * 640)) ** 0.5).to(dtype).cuda()
        yc = G.pf.conv_pf(G.pf.PF.from_nchw(xc), G.pf.pack_conv_weight(wc), b)
        assert _rel(yc.to_nchw(), F.conv2d(xc.float(), wc.float(), b.float(), padding=1)) <= TOL[dtype]
    finally:
        assert G.lib.gsw_mm_config(-1, 10) == 0


def test_mm_config_rejects_bad_values(G):
    assert G.lib.gsw_mm_config(64, -1) != 0 and G.lib.gsw_mm_config(0, 16) != 0 and G.lib.gsw_mm_config(-1, -1) == 0


# ---------------------------------------------------------------------------------------------------------------------------------
# whole models against the fp32 torch forward of the same module
# ---------------------------------------------------------------------------------------------------------------------------------
def _fp32_reference(U, m, x, t, c):
    """plain torch fp32 end to end: no own kernels anywhere (FUSED_KERNELS off), on a float copy of the module"""
    import copy
    mf = copy.deepcopy(m).float()
    U.FUSED_KERNELS = False
    try:
        return mf(x.float(), t, c.float())
    finally:
        U.FUSED_KERNELS = True


@pytest.mark.parametrize("B,rows", [(2, 0), (8, 256), (8, 128)], ids=["B2-auto", "B8-BM256", "B8-BM128"])
def test_full_sd21_unet_vs_fp32_torch(G, B, rows):
    """The eps model the loops call (extract.py:66-69: SD 2.1-base UNet, 865.9 M parameters, 64 x 64 lattice, t in {981, 1, ...}) on
    the hand-written kernels vs the fp32 torch forward of the same module.  Bound: 1e-2 of the output scale, absolute."""
    U = G.unet
    torch.manual_seed(0)
    m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
    assert sum(p.numel() for p in m.parameters()) == 865_910_724
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, 4, 64, 64, generator=g).cuda().half()
    c = torch.randn(B, 77, 1024, generator=g).cuda().half()
    t = torch.tensor(([981, 1, 500, 21, 741, 261, 901, 101] * 2)[:B]).cuda()
    U.FALLBACKS.clear()
    assert G.lib.gsw_mm_config(rows, -1) == 0
    try:
        with torch.no_grad():
            assert m._pf_ok(x)
            y = m(x, t, c)
    finally:
        assert G.lib.gsw_mm_config(0, -1) == 0
    assert U.FALLBACKS == {}, U.FALLBACKS
    with torch.no_grad():
        ref = _fp32_reference(U, m, x, t, c)
    scale = ref.abs().max().item()
    err = (y.float() - ref).abs().max().item()
    rms = ((y.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    print(f"full UNet B={B} rows={rows}: max|d|={err:.3e} scale={scale:.3f} rel-rms={rms:.3e}")
    assert torch.isfinite(y).all() and err <= 1e-2 * scale, (err, scale)
    assert rms <= 3e-3, rms


def test_unet_forward_is_bit_reproducible(G):
    """two forwards of the same input give the same bits (no float atomics, no order-dependent reductions left on the path)"""
    U = G.unet
    m = U.synthetic_init_(U.UNet2DCondition(block_out_channels=(320, 640, 1280, 1280), cross_attention_dim=1024, num_heads=(5, 10, 20, 20), head_dim=64), 0)
    m = m.cuda().half().eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 4, 64, 64, generator=g).cuda().half()
    c = torch.randn(4, 77, 1024, generator=g).cuda().half()
    t = torch.tensor([981, 1, 500, 21]).cuda()
    with torch.no_grad():
        y1 = m(x, t, c).clone()
        y2 = m(x, t, c).clone()
        y3 = m(x, t, c)
    assert torch.equal(y1, y2) and torch.equal(y1, y3)


def test_full_vae_vs_fp32_torch(G):
    """SD VAE (83.65 M parameters: 128 / 256 / 512 / 512) encode + decode on the PF kernels vs the fp32 torch forward; absolute bound"""
    V = G.vae
    v = V.synthetic_init_(V.AutoencoderKL(), 3).cuda().half().eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).cuda().half()
    z = torch.randn(2, 4, 32, 32, generator=g).cuda().half()
    V.FALLBACKS.clear()
    with torch.no_grad():
        e1, d1 = v.encode_mean(x), v.decode(z)
        assert V.FALLBACKS == {}, V.FALLBACKS
        import copy
        vf = copy.deepcopy(v).float()
        V.USE_PF = False
        try:
            er, dr = vf.encode_mean(x.float()), vf.decode(z.float())
        finally:
            V.USE_PF = True
    for a, r, name in ((e1, er, "encode"), (d1, dr, "decode")):
        err, scale = (a.float() - r).abs().max().item(), max(1.0, r.abs().max().item())
        print(f"full VAE {name}: max|d|={err:.3e} scale={scale:.3f}")
        assert err <= 1e-2 * scale, (name, err, scale)
