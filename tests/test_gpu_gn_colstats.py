"""GPU: GroupNorm statistics from the producing launch's column records (GswMmExtras.colstats_* -> gsw_groupnorm_pf_cs) against the separate
statistics pass and against fp32 torch GroupNorm of the same tensor: every producer (3x3 / stride-2 / 1x1 convolution, the three-segment resnet
launch, the sub-pixel upsampler's four parity launches, the token scatter), both tile heights, images smaller than a tile (8 x 8: four images per
256-row tile), the channel concatenation of two producers (a skip connection), and the fall-back when a launch cannot produce records."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, _native
    old, old_sk = pf.GN_FUSED_MAX_WGS, pf.SPLITK_MAX
    pf.GN_FUSED_MAX_WGS = 0          # this module is about the column records: small batches would otherwise take the one-launch GroupNorm (tests/test_gpu_small.py)
    pf.SPLITK_MAX = 1                # ... and a launch the plan splits writes none (its reduce kernel runs the epilogue): pinned off here, covered by
    yield types.SimpleNamespace(pf=pf, lib=_native.lib())          # test_a_split_producer_writes_no_records_and_groupnorm_takes_the_separate_pass below
    pf.GN_FUSED_MAX_WGS, pf.SPLITK_MAX = old, old_sk


@pytest.fixture(params=[0, 128, 256], ids=["auto", "BM128", "BM256"])
def tile_rows(request, G):
    assert G.lib.gsw_mm_config(request.param, -1) == 0
    yield request.param
    assert G.lib.gsw_mm_config(0, -1) == 0


def _gn_ref(x_nchw, gamma, beta, groups, eps, act):
    y = F.group_norm(x_nchw.float(), groups, gamma.float(), beta.float(), eps)
    return F.silu(y) if act else y


def _check_records(st, y_nchw):
    """the records themselves: folded over blocks they are the per-image, per-channel sums of the stored tensor"""
    B, C, H, W = y_nchw.shape
    rec = st.buf.view(st.npar, st.blocks, 2, C // 2).double()                        # [parity][block][sums | sums of squares][column pair]
    pix = H * W if st.npar == 1 else (H // 2) * (W // 2)
    bpi = pix // st.rows
    tot = rec[:, : B * bpi].reshape(st.npar, B, bpi, 2, C // 2).sum(dim=(0, 2))      # [B, 2, C / 2]
    yf = y_nchw.double().view(B, C // 2, 2, H, W)
    assert torch.allclose(tot[:, 0], yf.sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(tot[:, 1], (yf * yf).sum(dim=(2, 3, 4)), rtol=1e-4, atol=1e-2)


def _gn_both(G, y, gamma, beta, groups, eps, act, x2=None):
    pf = G.pf
    assert y.stats is not None and pf._stats_usable(y)
    a = pf.groupnorm_pf2(y, x2, gamma, beta, groups, eps, act=act)
    prev = pf.FUSE_GN_STATS
    pf.FUSE_GN_STATS = False
    try:
        b = pf.groupnorm_pf2(y, x2, gamma, beta, groups, eps, act=act)
    finally:
        pf.FUSE_GN_STATS = prev
    return a, b


@pytest.mark.parametrize("B,C,N,H,W,ks,stride", [(8, 320, 320, 32, 32, 3, 1), (84, 1280, 640, 8, 8, 3, 1), (40, 640, 1280, 32, 32, 3, 2), (30, 640, 320, 16, 16, 1, 1),
                                                  (2, 128, 256, 64, 64, 3, 1), (3, 320, 320, 32, 32, 3, 1)])
def test_conv_records_and_groupnorm(G, tile_rows, B, C, N, H, W, ks, stride):
    dt = torch.float16
    g = torch.Generator().manual_seed(B + C + N + H)
    x = torch.randn(B, C, H, W, generator=g).to(dt).cuda()
    w = (torch.randn(N, C, ks, ks, generator=g) * (1.0 / (C * ks * ks)) ** 0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    rb = torch.randn(B, N, generator=g).to(dt).cuda()
    P = G.pf.PF.from_nchw
    G.pf.LAUNCH_LOG = log = []
    try:
        y = G.pf.conv_pf(P(x), G.pf.pack_conv_weight(w), b, ksize=ks, stride=stride, rowbias=rb)
    finally:
        G.pf.LAUNCH_LOG = None
    if y.B * (y.H + 2) * (y.W + 2) <= 8192 and stride == 1:
        assert y.stats is None           # small tensors enumerate all padded rows (no records): GroupNorm takes the separate pass
        return
    if any(e.splits > 1 for e in log):   # a launch the plan splits (84 images at 8 x 8: 84 tiles of 256 rows -> 3 x 84 workgroups) writes no records either:
        assert y.stats is None           # its reduce kernel runs the epilogue; GroupNorm takes the separate pass, checked below like the record path
    else:
        _check_records(y.stats, y.to_nchw())
    gamma, beta = torch.randn(N, generator=g).to(dt).cuda(), torch.randn(N, generator=g).to(dt).cuda()
    a, bb = _gn_both(G, y, gamma, beta, 32, 1e-5, True)
    ref = _gn_ref(y.to_nchw(), gamma, beta, 32, 1e-5, True)
    assert (a.to_nchw().float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    assert (a.to_nchw().float() - bb.to_nchw().float()).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    assert a.grid[:, 0].abs().max() == 0 and a.grid[:, :, -1].abs().max() == 0


def test_resnet_tail_upsampler_and_concatenation(G, tile_rows):
    dt = torch.float16
    g = torch.Generator().manual_seed(11)
    rnd = lambda *s: torch.randn(*s, generator=g)
    B, C, N, H, W = 28, 640, 640, 16, 16
    P = G.pf.PF.from_nchw
    x, x1, x2 = (rnd(B, C, H, W).to(dt).cuda() for _ in range(3))
    w3 = (rnd(N, C, 3, 3) * (1.0 / (9 * C)) ** 0.5).to(dt).cuda()
    w1 = (rnd(N, 2 * C) * (1.0 / (2 * C)) ** 0.5).to(dt).cuda()
    bias = rnd(N).to(dt).cuda()
    tail = G.pf.conv3x3_res_pf(P(x), torch.cat([G.pf.pack_conv_weight(w3), w1], dim=1).contiguous(), bias, x1=P(x1), x2=P(x2))
    _check_records(tail.stats, tail.to_nchw())
    # sub-pixel upsampler: four parity launches, records over the low-resolution pixels
    wu = (rnd(N, N, 3, 3) * (1.0 / (9 * N)) ** 0.5).to(dt).cuda()
    up = G.pf.conv_up2x_pf(tail, G.pf.pack_upsample_weight(wu), bias)
    assert up.stats is not None and up.stats.npar == 4
    _check_records(up.stats, up.to_nchw())
    # a skip tensor at the upsampled resolution from a plain convolution; GroupNorm over the concatenation [up | skip] with 32 groups of
    # (640 + 320) / 32 = 30 channels: groups straddle the boundary between the two tensors
    xs = rnd(B, 320, 2 * H, 2 * W).to(dt).cuda()
    ws = (rnd(320, 320, 3, 3) * (1.0 / (9 * 320)) ** 0.5).to(dt).cuda()
    skip = G.pf.conv_pf(P(xs), G.pf.pack_conv_weight(ws), None)
    Ct = N + 320
    gamma, beta = rnd(Ct).to(dt).cuda(), rnd(Ct).to(dt).cuda()
    a, bb = _gn_both(G, up, gamma, beta, 32, 1e-5, True, x2=skip)
    ref = _gn_ref(torch.cat([up.to_nchw(), skip.to_nchw()], dim=1), gamma, beta, 32, 1e-5, True)
    assert (a.to_nchw().float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    assert (a.to_nchw().float() - bb.to_nchw().float()).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_token_scatter_replaces_the_records_and_tokens_output(G, tile_rows):
    dt = torch.float16
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 8, 32, 32, 320
    base = torch.randn(B, C, H, W, generator=g).to(dt).cuda()
    wc = (torch.randn(C, C, 3, 3, generator=g) * (1.0 / (9 * C)) ** 0.5).to(dt).cuda()
    x = G.pf.conv_pf(G.pf.PF.from_nchw(base), G.pf.pack_conv_weight(wc), None)
    old = x.stats
    tok = torch.randn(B, H * W, C, generator=g).to(dt).cuda()
    w = (torch.randn(C, C, generator=g) * C ** -0.5).to(dt).cuda()
    b = torch.randn(C, generator=g).to(dt).cuda()
    G.pf.gemm(tok, w, b, resid=x.rows, mode="tok2pf", tokens=H * W, width=W, out=x.rows, stats_for=x)
    assert x.stats is not None and x.stats is not old
    _check_records(x.stats, x.to_nchw())
    gamma, beta = torch.randn(C, generator=g).to(dt).cuda(), torch.randn(C, generator=g).to(dt).cuda()
    t_cs = G.pf.groupnorm_pf(x, gamma, beta, 32, 1e-6, act=False, tokens=True)
    ref = _gn_ref(x.to_nchw(), gamma, beta, 32, 1e-6, False).permute(0, 2, 3, 1).reshape(B, H * W, C)
    assert (t_cs.float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


def test_a_launch_off_the_engine_reports_no_records(G):
    """a convolution off the engine (64 output columns) writes no records and says so in its extras; nothing is armed for a later launch (ABI 0.5.0: there is no
    state a request could linger in)"""
    import ctypes as C
    from gswm_amd import _native as N
    from gswm_amd.codec import _dt
    buf = torch.zeros(1 << 16, dtype=torch.float32, device="cuda")
    x = G.pf.PF.from_nchw(torch.randn(2, 64, 16, 16, device="cuda").half())
    w = G.pf.pack_conv_weight(torch.randn(64, 64, 3, 3, device="cuda").half() * 0.05)
    y = G.pf.PF.empty(2, 16, 16, 64, torch.float16, "cuda")
    ex = N.GswMmExtras()
    ex.colstats_dev, ex.colstats_capacity = buf.data_ptr(), buf.numel()
    assert G.lib.gsw_conv_pf_ex(x.rows.data_ptr(), w.data_ptr(), None, None, 0, None, y.rows.data_ptr(), 2, 16, 16, 64, 64, 3, 1, 64, _dt(torch.float16), C.byref(ex), None) == 0
    assert ex.colstats_rows_per_block == 0 and ex.colstats_blocks == 0
    w2 = G.pf.pack_conv_weight(torch.randn(128, 64, 3, 3, device="cuda").half() * 0.05)
    y2 = G.pf.PF.empty(2, 16, 16, 128, torch.float16, "cuda")
    assert G.lib.gsw_conv_pf(x.rows.data_ptr(), w2.data_ptr(), None, None, 0, None, y2.rows.data_ptr(), 2, 16, 16, 64, 128, 3, 1, 64, _dt(torch.float16), None) == 0      # an engine launch without extras
    torch.cuda.synchronize()
    assert float(buf.abs().sum()) == 0.0
    ex.colstats_dev = buf.data_ptr() + 4
    assert G.lib.gsw_conv_pf_ex(x.rows.data_ptr(), w2.data_ptr(), None, None, 0, None, y2.rows.data_ptr(), 2, 16, 16, 64, 128, 3, 1, 64, _dt(torch.float16), C.byref(ex), None) != 0      # alignment
    ex.colstats_dev, ex.flags = buf.data_ptr(), 0x40
    assert G.lib.gsw_conv_pf_ex(x.rows.data_ptr(), w2.data_ptr(), None, None, 0, None, y2.rows.data_ptr(), 2, 16, 16, 64, 128, 3, 1, 64, _dt(torch.float16), C.byref(ex), None) == N.GSW_ERR_BAD_ARG      # unknown flag bits


def test_explicit_extras_report_what_the_launch_did(G):
    """GswMmExtras (ABI 0.4.0): the records requested and the split-K scratch travel with the launch, and the launch says whether it wrote records --
    a split-K launch drops the request VISIBLY (rows per block 0, splits > 1), the one-shot form returns GSW_WARN_NO_RECORDS for the same case"""
    import ctypes as C
    from gswm_amd import _native as N
    from gswm_amd.codec import _dt
    B, Cc, Nn, H, W = 1, 1280, 1280, 8, 8
    x = G.pf.PF.from_nchw(torch.randn(B, Cc, H, W, device="cuda").half())
    w = G.pf.pack_conv_weight((torch.randn(Nn, Cc, 3, 3, device="cuda") * 0.01).half())
    y = G.pf.PF.empty(B, H, W, Nn, torch.float16, "cuda")
    rec = torch.zeros(4 * 4 * Nn, dtype=torch.float32, device="cuda")
    ws = torch.empty(G.pf.SPLITK_BYTES, dtype=torch.uint8, device="cuda")
    ex = N.GswMmExtras()
    ex.colstats_dev, ex.colstats_capacity = rec.data_ptr(), rec.numel()
    ex.workspace_dev, ex.workspace_bytes, ex.max_splits = ws.data_ptr(), ws.numel(), 0
    assert G.lib.gsw_conv_pf_ex(x.rows.data_ptr(), w.data_ptr(), None, None, 0, None, y.rows.data_ptr(), B, H, W, Cc, Nn, 3, 1, Cc, _dt(torch.float16), C.byref(ex), None) == 0
    assert ex.splits > 1 and ex.colstats_rows_per_block == 0 and ex.colstats_blocks == 0          # one image's 8 x 8 level: split-K, no records
    y_split = y.to_nchw().clone()
    ex2 = N.GswMmExtras()                                                                          # no workspace: unsplit
    ex2.colstats_dev, ex2.colstats_capacity = rec.data_ptr(), rec.numel()
    assert G.lib.gsw_conv_pf_ex(x.rows.data_ptr(), w.data_ptr(), None, None, 0, None, y.rows.data_ptr(), B, H, W, Cc, Nn, 3, 1, Cc, _dt(torch.float16), C.byref(ex2), None) == 0
    assert ex2.splits == 1
    assert (y.to_nchw().float() - y_split.float()).abs().max().item() <= 2e-3 * y_split.float().abs().max().item()
    torch.cuda.synchronize()


def test_a_split_producer_writes_no_records_and_groupnorm_takes_the_separate_pass(G):
    """84 images at 8 x 8, 1280 -> 640: 84 tiles of 256 rows -- the plan splits K three ways (csrc/gswm_mm.hip: mm_plan); the reduce kernel writes no records"""
    dt = torch.float16
    g = torch.Generator().manual_seed(5)
    B, C, N, H, W = 84, 1280, 640, 8, 8
    x = torch.randn(B, C, H, W, generator=g).to(dt).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).to(dt).cuda()
    b = torch.randn(N, generator=g).to(dt).cuda()
    prev = G.pf.SPLITK_MAX
    G.pf.SPLITK_MAX = 0
    G.pf.LAUNCH_LOG = log = []
    try:
        y = G.pf.conv_pf(G.pf.PF.from_nchw(x), G.pf.pack_conv_weight(w), b)
    finally:
        G.pf.LAUNCH_LOG = None
        G.pf.SPLITK_MAX = prev
    assert any(e.splits > 1 for e in log) and y.stats is None
    gamma, beta = torch.randn(N, generator=g).to(dt).cuda(), torch.randn(N, generator=g).to(dt).cuda()
    a = G.pf.groupnorm_pf2(y, None, gamma, beta, 32, 1e-5, act=True)
    ref = _gn_ref(y.to_nchw(), gamma, beta, 32, 1e-5, True)
    assert (a.to_nchw().float() - ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,C,N,H,W", [(8, 320, 320, 32, 32), (20, 640, 1280, 16, 16)])
def test_gn_only_convolution_skips_the_border_and_groupnorm_does_not_care(G, B, C, N, H, W):
    """conv_pf(gn_only=True) (a resnet's conv1 -> norm2): the launch that wrote column records leaves the output's border unwritten (border_valid False, one launch
    less); the record-fed GroupNorm gives the same bits as behind a zero-bordered producer, and a GroupNorm that has to take the statistics pass zeroes the border first."""
    pf = G.pf
    g = torch.Generator().manual_seed(B + C + N)
    x = torch.randn(B, C, H, W, generator=g).half().cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    gamma, beta = torch.randn(N, generator=g).half().cuda(), torch.randn(N, generator=g).half().cuda()
    X, Wp = pf.PF.from_nchw(x), pf.pack_conv_weight(w)
    y0 = pf.conv_pf(X, Wp, b)
    assert y0.border_valid and y0.stats is not None
    ref = pf.groupnorm_pf(y0, gamma, beta, 32, 1e-5)
    # poison fresh allocations so that an unwritten border is not accidentally zero
    junk = torch.full((y0.buf.numel(),), float("nan"), dtype=torch.float16, device="cuda")
    del junk
    y1 = pf.conv_pf(X, Wp, b, gn_only=True)
    assert y1.stats is not None and not y1.border_valid
    assert torch.equal(y1.interior, y0.interior)
    out = pf.groupnorm_pf(y1, gamma, beta, 32, 1e-5)
    assert torch.equal(out.rows, ref.rows)                            # interior AND (zero) border of the GroupNorm output
    # statistics-pass fall-back: drop the records -> the border is zeroed before the pass reads it
    y1.stats = None
    out2 = pf.groupnorm_pf(y1, gamma, beta, 32, 1e-5)
    assert y1.border_valid and y1.grid[:, 0].abs().max() == 0 and y1.grid[:, :, -1].abs().max() == 0
    r = _gn_ref(y0.to_nchw(), gamma, beta, 32, 1e-5, True)
    assert (out2.to_nchw().float() - r).abs().max().item() <= 2e-2 * max(1.0, r.abs().max().item())
