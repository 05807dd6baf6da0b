"""GPU: self-attention's q | k | v projections from one pass over the tokens (gsw_gemm_qkv: column tiles below N_rows take the dense-row epilogue, the
others the transposed one with swapped MFMA operands, decided per tile) against the two separate launches it replaces -- bit for bit, since every
output element sees the same K order -- and against fp32 torch; both tile heights, ragged M, the split-K form, and the UNet's attention through it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import types
    import gswm_amd
    from gswm_amd import pf, unet, _native
    old = pf.SMALL_GEMM_MAX_ROWS
    pf.SMALL_GEMM_MAX_ROWS = 0          # the bit-for-bit comparisons of this module are engine against engine (at <= 128 rows pf.gemm would take gsw_gemm_small)
    yield types.SimpleNamespace(pf=pf, unet=unet, lib=_native.lib())
    pf.SMALL_GEMM_MAX_ROWS = old


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows", [0, 128, 256])
@pytest.mark.parametrize("B,S,K,inner", [(3, 256, 1280, 1280), (2, 4096, 320, 320), (70, 64, 1280, 1280), (5, 1024, 640, 640), (1, 64, 1280, 1280), (1, 4096, 320, 320)])
def test_qkv_one_pass_equals_the_two_launches(G, dtype, rows, B, S, K, inner):
    g = torch.Generator().manual_seed(B + S + K)
    x = torch.randn(B, S, K, generator=g).to(dtype).cuda()
    wq, wk, wv = ((torch.randn(inner, K, generator=g) * K ** -0.5).to(dtype).cuda() for _ in range(3))
    wqkv = torch.cat([wq, wk, wv], dim=0).contiguous()
    assert G.lib.gsw_mm_config(rows, -1) == 0
    try:
        qk, vt = G.pf.gemm_qkv(x, wqkv, 2 * inner)
        qk_ref = G.pf.gemm(x, torch.cat([wq, wk], dim=0).contiguous(), None)
        vt_ref = G.pf.gemm(x, wv, None, mode="trans", tokens=S)
    finally:
        assert G.lib.gsw_mm_config(0, -1) == 0
    assert qk.shape == (B, S, 2 * inner) and vt.shape == (B, inner, S)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    f = x.float()
    assert ((qk.float() - f @ torch.cat([wq, wk], 0).float().T).abs().max() / (f @ wq.float().T).abs().max()).item() <= tol
    assert ((vt.float() - (f @ wv.float().T).transpose(1, 2)).abs().max() / (f @ wv.float().T).abs().max()).item() <= tol
    # same tiling, same K order per element: identical bits (when the launches tile alike: the one-pass launch has 1.5x the column tiles, so
    # the automatic choice between 128- and 256-row tiles / split-K can differ -- then the tolerance above is the statement)
    if rows:
        prev = G.pf.SPLITK_MAX
        G.pf.SPLITK_MAX = 1
        try:
            assert G.lib.gsw_mm_config(rows, -1) == 0
            qk1, vt1 = G.pf.gemm_qkv(x, wqkv, 2 * inner)
            qk_r1 = G.pf.gemm(x, torch.cat([wq, wk], dim=0).contiguous(), None)
            vt_r1 = G.pf.gemm(x, wv, None, mode="trans", tokens=S)
        finally:
            G.pf.SPLITK_MAX = prev
            assert G.lib.gsw_mm_config(0, -1) == 0
        assert torch.equal(qk1, qk_r1) and torch.equal(vt1, vt_r1)


def test_qkv_with_bias_and_argument_checks(G):
    dtype = torch.float16
    g = torch.Generator().manual_seed(1)
    B, S, K, inner = 2, 128, 320, 320
    x = torch.randn(B, S, K, generator=g).to(dtype).cuda()
    w = (torch.randn(3 * inner, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(3 * inner, generator=g).to(dtype).cuda()
    qk, vt = G.pf.gemm_qkv(x, w, 2 * inner, b)
    ref = x.float() @ w.float().T + b.float()
    assert (qk.float() - ref[..., : 2 * inner]).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (vt.float() - ref[..., 2 * inner:].transpose(1, 2)).abs().max().item() <= 2e-3 * ref.abs().max().item()
    with pytest.raises(ValueError):
        G.pf.gemm_qkv(x, w, 2 * inner + 8)                     # a column tile must be of one kind
    p = x.data_ptr()
    assert G.lib.gsw_gemm_qkv(p, p, None, p, p, 256, 320, 600, 960, 128, 1, None) != 0
    assert G.lib.gsw_gemm_qkv(p, p, None, p, p, 256, 320, 640, 960, 100, 1, None) != 0       # S % 8


def test_attention_block_through_the_one_pass_projection(G):
    """unet.Attention (self-attention) with FUSED_QKV on / off: identical output bits at the UNet's 32 x 32 level"""
    U = G.unet
    torch.manual_seed(0)
    att = U.Attention(640, 640, 10, 64).cuda().half().eval()
    x = torch.randn(4, 1024, 640, device="cuda").half()
    r = torch.randn(4, 1024, 640, device="cuda").half()
    prev = G.pf.SPLITK_MAX
    G.pf.SPLITK_MAX = 1
    try:
        assert G.lib.gsw_mm_config(256, -1) == 0
        with torch.no_grad():
            y0 = att(x, resid=r)
            U.FUSED_QKV = True
            try:
                y1 = att(x, resid=r)
            finally:
                U.FUSED_QKV = False
    finally:
        G.pf.SPLITK_MAX = prev
        assert G.lib.gsw_mm_config(0, -1) == 0
    assert torch.equal(y0, y1)
