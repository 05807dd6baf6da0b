"""GPU: split-K launches of the matmul engine (GswMmExtras.workspace_*; csrc/gswm_mm.hip EPI 4 + gsw_mm_reduce_kernel) against fp32 torch and against the
unsplit launch of the same operands -- every epilogue mode, K ranges that start inside a tap run / cross a segment boundary, ragged M and N,
forced split counts that do not divide the stage count, and the automatic policy on the small-batch shapes of the eps model (one image's
8 x 8 and 16 x 16 levels: the reference's one-latent-per-call regime, extract.py:112-117)."""
import contextlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, unet, _native
    old = pf.SMALL_GEMM_MAX_ROWS
    pf.SMALL_GEMM_MAX_ROWS = 0          # this module tests the matmul ENGINE (at <= 128 rows pf.gemm would otherwise take gsw_gemm_small: tests/test_gpu_small.py)
    yield types.SimpleNamespace(pf=pf, unet=unet, lib=_native.lib())
    pf.SMALL_GEMM_MAX_ROWS = old


@contextlib.contextmanager
def splits(G, k):
    """k = 1: never split; k > 1: force k-way (where K allows); 0: automatic"""
    prev = G.pf.SPLITK_MAX
    G.pf.SPLITK_MAX = k
    try:
        yield
    finally:
        G.pf.SPLITK_MAX = prev


def _rel(y, ref):
    return (y.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-6)


TOL = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 3, 7, 16])
@pytest.mark.parametrize("M,K,N", [(64, 1280, 1280), (1, 1280, 640), (200, 5120, 1280), (256, 1280, 328), (1000, 640, 640)])
def test_dense_rows_split_vs_fp32_and_unsplit(G, dtype, k, M, K, N):
    g = torch.Generator().manual_seed(M + K + N + k)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    ref = x.float() @ w.float().T + b.float()
    with splits(G, 1):
        y1, y1r = G.pf.gemm(x, w, b), G.pf.gemm(x, w, b, resid=r)
    with splits(G, k):
        yk, ykr = G.pf.gemm(x, w, b), G.pf.gemm(x, w, b, resid=r)
        yk2 = G.pf.gemm(x, w, b)
    assert torch.equal(yk, yk2)                                    # fixed summation order: deterministic
    assert _rel(yk, ref) <= TOL[dtype] and _rel(ykr, ref + r.float()) <= 2 * TOL[dtype]
    assert _rel(yk, y1) <= TOL[dtype] and _rel(ykr, y1r) <= 2 * TOL[dtype]


@pytest.mark.parametrize("k", [2, 5])
def test_geglu_trans_tok2pf_split(G, k):
    dtype = torch.float16
    g = torch.Generator().manual_seed(k)
    # GEGLU
    M, K, I = 256, 1280, 640
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(2 * I, generator=g).to(dtype).cuda()
    wp, bp = G.pf.pack_geglu_weight(w, b)
    proj = x.float() @ w.float().T + b.float()
    ref = proj[:, :I] * F.gelu(proj[:, I:])
    with splits(G, k):
        y = G.pf.gemm(x, wp, bp, mode="geglu")
    with splits(G, 1):
        y1 = G.pf.gemm(x, wp, bp, mode="geglu")
    assert _rel(y, ref) <= 3e-3 and _rel(y, y1) <= 2e-3
    # transposed value projection
    Bn, S, K2, N2 = 3, 64, 1280, 1280
    xs = torch.randn(Bn, S, K2, generator=g).to(dtype).cuda()
    w2 = (torch.randn(N2, K2, generator=g) * K2 ** -0.5).to(dtype).cuda()
    b2 = torch.randn(N2, generator=g).to(dtype).cuda()
    refT = (xs.float() @ w2.float().T + b2.float()).transpose(1, 2)
    with splits(G, k):
        yT = G.pf.gemm(xs, w2, b2, mode="trans", tokens=S)
    assert yT.shape == (Bn, N2, S) and _rel(yT, refT) <= 2e-3
    # tokens -> PF interior rows, residual = the target itself
    Bn, H, W, C = 2, 8, 8, 1280
    tok = torch.randn(Bn, H * W, C, generator=g).to(dtype).cuda()
    base = torch.randn(Bn, C, H, W, generator=g).to(dtype).cuda()
    w3 = (torch.randn(C, C, generator=g) * C ** -0.5).to(dtype).cuda()
    b3 = torch.randn(C, generator=g).to(dtype).cuda()
    refP = base.float() + (tok.float() @ w3.float().T + b3.float()).view(Bn, H, W, C).permute(0, 3, 1, 2)
    pfx = G.pf.PF.from_nchw(base)
    with splits(G, k):
        G.pf.gemm(tok, w3, b3, resid=pfx.rows, mode="tok2pf", tokens=H * W, width=W, out=pfx.rows)
    assert _rel(pfx.to_nchw(), refP) <= 4e-3
    assert pfx.grid[:, 0].abs().max() == 0 and pfx.grid[:, :, -1].abs().max() == 0


# (B, C_in, C_out, H, W, ksize, stride): one image's deep levels and friends
CONVS = [(1, 1280, 1280, 8, 8, 3, 1), (2, 2560, 1280, 8, 8, 3, 1), (1, 1280, 1280, 16, 16, 3, 1), (1, 640, 1280, 16, 16, 1, 1), (3, 640, 640, 32, 32, 3, 2),
         (1, 128, 136, 10, 6, 3, 1)]


@pytest.mark.parametrize("k", [0, 2, 4, 9, 16])
@pytest.mark.parametrize("B,C,N,H,W,ks,stride", CONVS)
def test_conv_split_vs_fp32(G, k, B, C, N, H, W, ks, stride):
    dtype = torch.float16
    Ho, Wo = H // stride, W // stride
    g = torch.Generator().manual_seed(C + N + H + ks + k)
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(N, C, ks, ks, generator=g) * (1.0 / (C * ks * ks)) ** 0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    rb = torch.randn(B, N, generator=g).to(dtype).cuda()
    res = torch.randn(B, N, Ho, Wo, generator=g).to(dtype).cuda()
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=ks // 2, stride=stride) + rb.float()[:, :, None, None] + res.float()
    P = G.pf.PF.from_nchw
    with splits(G, k):
        y = G.pf.conv_pf(P(x), G.pf.pack_conv_weight(w), b, ksize=ks, stride=stride, rowbias=rb, resid=P(res))
    assert _rel(y.to_nchw(), ref) <= 2e-3
    gr = y.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


@pytest.mark.parametrize("k", [0, 3, 8, 13])
def test_three_segment_resnet_launch_split(G, k):
    """conv2 + conv_shortcut over cat(x1, x2): the K ranges of the splits start inside the 3x3 segment, at its end, and inside the 1x1 segments"""
    dtype = torch.float16
    B, C, N, C1, C2, H, W = 1, 1280, 1280, 1280, 640, 8, 8
    g = torch.Generator().manual_seed(k)
    rnd = lambda *s: torch.randn(*s, generator=g)
    x = rnd(B, C, H, W).to(dtype).cuda()
    x1, x2 = rnd(B, C1, H, W).to(dtype).cuda(), rnd(B, C2, H, W).to(dtype).cuda()
    w3 = (rnd(N, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).to(dtype).cuda()
    w1 = (rnd(N, C1 + C2) * (1.0 / (C1 + C2)) ** 0.5).to(dtype).cuda()
    b = rnd(N).to(dtype).cuda()
    ref = F.conv2d(x.float(), w3.float(), b.float(), padding=1) + F.conv2d(torch.cat([x1, x2], 1).float(), w1.float()[:, :, None, None])
    P = G.pf.PF.from_nchw
    with splits(G, k):
        y = G.pf.conv3x3_res_pf(P(x), torch.cat([G.pf.pack_conv_weight(w3), w1], dim=1).contiguous(), b, x1=P(x1), x2=P(x2))
    assert _rel(y.to_nchw(), ref) <= 2e-3


@pytest.mark.parametrize("k", [0, 2, 6])
def test_up2x_split(G, k):
    dtype = torch.float16
    B, H, W, C, N = 1, 8, 8, 1280, 1280
    g = torch.Generator().manual_seed(k + 40)
    x = torch.randn(B, C, H, W, generator=g).to(dtype).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dtype).cuda()
    b = (0.1 * torch.randn(N, generator=g)).to(dtype).cuda()
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    with splits(G, k):
        y = G.pf.conv_up2x_pf(G.pf.PF.from_nchw(x), G.pf.pack_upsample_weight(w), b)
    assert _rel(y.to_nchw(), ref) <= 4e-3
    gr = y.grid
    assert gr[:, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0


def test_workspace_abi_validation_and_no_workspace_means_unsplit(G):
    lib = G.lib
    import ctypes as C
    from gswm_amd import _native as N
    buf = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    a = torch.randn(64, 320, device="cuda").half()
    wt = torch.randn(320, 320, device="cuda").half()
    o = torch.empty(64, 320, device="cuda").half()

    def launch(ex):
        return lib.gsw_gemm_ex(a.data_ptr(), 320, wt.data_ptr(), 320, None, None, 320, o.data_ptr(), 320, 64, 320, 320, 0, 0, 0, 1, C.byref(ex), None)
    for ws, nbytes, ms in ((None, -1, 0), (None, 0, 65), (buf.data_ptr() + 8, 1024, 0)):          # negative size, too many splits, 16-byte alignment
        ex = N.GswMmExtras()
        ex.workspace_dev, ex.workspace_bytes, ex.max_splits = ws, nbytes, ms
        assert launch(ex) != 0
    # a workspace too small for the split a launch would take: the launch runs unsplit and still gives the right answer
    x = torch.randn(64, 1280, device="cuda").half()
    w = (torch.randn(1280, 1280, device="cuda") * 1280 ** -0.5).half()
    ref = x.float() @ w.float().T
    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    with G.pf.splitk_workspace(small):
        y = G.pf.gemm(x, w, None)
    assert _rel(y, ref) <= 2e-3


def test_small_batch_unet_forward_split_vs_unsplit_vs_fp32(G):
    """One image through the SD 2.1-shaped UNet: the automatic split-K policy against the unsplit engine and the fp32 torch forward."""
    U = G.unet
    m = U.synthetic_init_(U.UNet2DCondition(), 0).cuda().half().eval()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 4, 64, 64, generator=g).cuda().half()
    c = torch.randn(1, 77, 1024, generator=g).cuda().half()
    t = torch.full((), 481, dtype=torch.int64, device="cuda")
    U.FALLBACKS.clear()
    with torch.no_grad():
        with splits(G, 1):
            y1 = m(x, t, c)
        with splits(G, 0):
            y0 = m(x, t, c)
            y0b = m(x, t, c)
        assert torch.equal(y0, y0b)
        import copy
        mf = copy.deepcopy(m).float()
        U.FUSED_KERNELS = False
        try:
            ref = mf(x.float(), t, c.float())
        finally:
            U.FUSED_KERNELS = True
    scale = ref.abs().max().item()
    assert (y0.float() - ref).abs().max().item() <= 1e-2 * scale and (y1.float() - ref).abs().max().item() <= 1e-2 * scale
    assert U.FALLBACKS == {}, U.FALLBACKS


# ---------------------------------------------------------------------------------------------------------------------
# 256-row tiles in split launches (round 4): forced through gsw_mm_config, and the automatic policy on the shapes it was added for
# ---------------------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def tile_rows(G, rows):
    import ctypes as C
    tr, sm = C.c_int(0), C.c_int(0)
    assert G.lib.gsw_mm_get_config(C.byref(tr), C.byref(sm)) == 0
    assert G.lib.gsw_mm_config(rows, sm.value) == 0
    try:
        yield
    finally:
        assert G.lib.gsw_mm_config(tr.value, sm.value) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 5])
def test_256_row_split_tiles_every_mode(G, dtype, k):
    g = torch.Generator().manual_seed(7 * k)
    rnd = lambda *s: torch.randn(*s, generator=g)
    tol = TOL[dtype]
    with tile_rows(G, 256):
        # dense rows (+ residual), ragged M and N
        M, K, N = 1000, 5120, 1288
        x, w, b, r = rnd(M, K).to(dtype).cuda(), (rnd(N, K) * K ** -0.5).to(dtype).cuda(), rnd(N).to(dtype).cuda(), rnd(M, N).to(dtype).cuda()
        ref = x.float() @ w.float().T + b.float() + r.float()
        G.pf.LAUNCH_LOG = log = []
        try:
            with splits(G, k):
                y, y2 = G.pf.gemm(x, w, b, resid=r), G.pf.gemm(x, w, b, resid=r)
        finally:
            G.pf.LAUNCH_LOG = None
        assert [e.splits for e in log] == [k, k]
        assert torch.equal(y, y2) and _rel(y, ref) <= 2 * tol
        with splits(G, 1):
            y1 = G.pf.gemm(x, w, b, resid=r)
        assert _rel(y, y1) <= 2 * tol
        # GEGLU
        M, K, I = 512, 1280, 640
        x, w, b = rnd(M, K).to(dtype).cuda(), (rnd(2 * I, K) * K ** -0.5).to(dtype).cuda(), rnd(2 * I).to(dtype).cuda()
        wp, bp = G.pf.pack_geglu_weight(w, b)
        proj = x.float() @ w.float().T + b.float()
        with splits(G, k):
            yg = G.pf.gemm(x, wp, bp, mode="geglu")
        assert _rel(yg, proj[:, :I] * F.gelu(proj[:, I:])) <= 2 * tol
        # transposed value projection
        Bn, S, K2, N2 = 3, 256, 1280, 1280
        xs, w2, b2 = rnd(Bn, S, K2).to(dtype).cuda(), (rnd(N2, K2) * K2 ** -0.5).to(dtype).cuda(), rnd(N2).to(dtype).cuda()
        with splits(G, k):
            yT = G.pf.gemm(xs, w2, b2, mode="trans", tokens=S)
        assert _rel(yT, (xs.float() @ w2.float().T + b2.float()).transpose(1, 2)) <= tol
        # convolution with row bias and residual; the resnet's three-segment launch; the sub-pixel upsampler
        B, C, Nn, H, W = 3, 1280, 1280, 16, 16
        xc, wc, bc = rnd(B, C, H, W).to(dtype).cuda(), (rnd(Nn, C, 3, 3) * (9 * C) ** -0.5).to(dtype).cuda(), rnd(Nn).to(dtype).cuda()
        rb, res = rnd(B, Nn).to(dtype).cuda(), rnd(B, Nn, H, W).to(dtype).cuda()
        refc = F.conv2d(xc.float(), wc.float(), bc.float(), padding=1) + rb.float()[:, :, None, None] + res.float()
        P = G.pf.PF.from_nchw
        with splits(G, k):
            yc = G.pf.conv_pf(P(xc), G.pf.pack_conv_weight(wc), bc, rowbias=rb, resid=P(res))
        assert _rel(yc.to_nchw(), refc) <= 2 * tol
        gr = yc.grid
        assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0
        x1, x2, w1 = rnd(B, 1280, H, W).to(dtype).cuda(), rnd(B, 640, H, W).to(dtype).cuda(), (rnd(Nn, 1920) * 1920 ** -0.5).to(dtype).cuda()
        ref3 = F.conv2d(xc.float(), wc.float(), bc.float(), padding=1) + F.conv2d(torch.cat([x1, x2], 1).float(), w1.float()[:, :, None, None])
        with splits(G, k):
            y3 = G.pf.conv3x3_res_pf(P(xc), torch.cat([G.pf.pack_conv_weight(wc), w1], dim=1).contiguous(), bc, x1=P(x1), x2=P(x2))
        assert _rel(y3.to_nchw(), ref3) <= 2 * tol
        refu = F.conv2d(F.interpolate(xc.float(), scale_factor=2.0, mode="nearest"), wc.float(), bc.float(), padding=1)
        with splits(G, k):
            yu = G.pf.conv_up2x_pf(P(xc), G.pf.pack_upsample_weight(wc), bc)
        assert _rel(yu.to_nchw(), refu) <= 3 * tol


def test_automatic_policy_takes_256_row_tiles_where_half_the_cus_would_idle(G):
    """16 images at 16 x 16 (the third level at batch 8 with guidance) and 64 at 8 x 8 (the deepest level of the inversion half at batch 64): 4096 output rows x
    1280 columns = 128 tiles of 256 rows = half the chip.  With 128-row tiles such a launch cannot split at all (256 tiles); from K = 17280 up the plan takes
    2 x 128 workgroups of 256-row tiles (and the interior row enumeration that makes them 128), below it 256 unsplit 128-row tiles (csrc/gswm_mm.hip: mm_plan)."""
    dtype = torch.float16
    g = torch.Generator().manual_seed(11)
    for (B, H, C, want) in ((16, 16, 1920, 2), (64, 8, 2560, 2), (16, 16, 1280, 1)):
        x = torch.randn(B, C, H, H, generator=g).to(dtype).cuda()
        w = (torch.randn(1280, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dtype).cuda()
        b = torch.randn(1280, generator=g).to(dtype).cuda()
        ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
        G.pf.LAUNCH_LOG = log = []
        try:
            y = G.pf.conv_pf(G.pf.PF.from_nchw(x), G.pf.pack_conv_weight(w), b)
        finally:
            G.pf.LAUNCH_LOG = None
        assert [e.splits for e in log] == [want], (B, H, C, [e.splits for e in log])
        assert _rel(y.to_nchw(), ref) <= 2e-3
        gr = y.grid
        assert gr[:, 0].abs().max() == 0 and gr[:, -1].abs().max() == 0 and gr[:, :, 0].abs().max() == 0 and gr[:, :, -1].abs().max() == 0
    # ... and leaves launches alone that fill the chip: 128 images at 8 x 8 (256 tiles of 256 rows)
    x = torch.randn(128, 1280, 8, 8, generator=g).to(dtype).cuda()
    w = (torch.randn(1280, 1280, 3, 3, generator=g) * (9 * 1280) ** -0.5).to(dtype).cuda()
    G.pf.LAUNCH_LOG = log = []
    try:
        G.pf.conv_pf(G.pf.PF.from_nchw(x), G.pf.pack_conv_weight(w), b)
    finally:
        G.pf.LAUNCH_LOG = None
    assert [e.splits for e in log] == [1]
