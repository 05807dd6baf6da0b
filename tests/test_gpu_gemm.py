"""GPU: the matmul engine (csrc/gswm_mm.hip, C ABI gsw_gemm) against torch fp32 references: every transformer-linear shape class of
the eps model (diffusers BasicTransformerBlock behind extract.py:66-69), every epilogue, ragged M, tiny and tail grids."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import gswm_amd
    from gswm_amd import pf
    old = pf.SMALL_GEMM_MAX_ROWS
    pf.SMALL_GEMM_MAX_ROWS = 0          # this module tests the matmul ENGINE (at <= 128 rows pf.gemm would otherwise take gsw_gemm_small: tests/test_gpu_small.py)
    yield pf
    pf.SMALL_GEMM_MAX_ROWS = old


def _tol(dtype):
    return 2e-3 if dtype == torch.float16 else 1.6e-2          # one rounding of the storage dtype (two with a residual) at the output magnitude


def _mk(M, K, N, dtype, seed, resid=False):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda() if resid else None
    return x, w, b, r


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(256, 64, 160), (256, 128, 320), (8, 320, 320), (513, 192, 160), (1000, 320, 320), (4096, 320, 960), (777, 640, 640), (2048, 1280, 320),
                                   (300, 1280, 1280), (16384, 320, 320), (70000, 320, 320), (12800, 640, 1920), (128, 1024, 1280), (64, 1280, 20160), (70000, 128, 320)])
def test_gemm_plain_vs_torch_fp32(P, dtype, M, K, N):
    x, w, b, r = _mk(M, K, N, dtype, M + K + N, resid=True)
    ref = x.float() @ w.float().T + b.float()
    y = P.gemm(x, w, b)
    assert y.shape == (M, N)
    assert (y.float() - ref).abs().max().item() <= _tol(dtype) * ref.abs().max().item()
    y2 = P.gemm(x, w, None, resid=r)
    ref2 = x.float() @ w.float().T + r.float()
    assert (y2.float() - ref2).abs().max().item() <= 2 * _tol(dtype) * ref2.abs().max().item()


def test_gemm_detects_transposes(P):
    """identity activations against an asymmetric weight matrix: a swapped operand or output index cannot pass"""
    K = N = 320
    x = torch.eye(K, dtype=torch.float16).cuda()
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 / 251.0).half().cuda()
    y = P.gemm(x, w, None)
    assert torch.equal(y, w.T.contiguous())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,I", [(512, 320, 1280), (1000, 640, 2560), (256, 1280, 5120), (9000, 320, 1280), (300, 128, 160), (520, 192, 320), (33000, 192, 1280), (40000, 128, 320)])
def test_gemm_geglu_vs_torch(P, dtype, M, K, I):
    g = torch.Generator().manual_seed(M + I)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = (0.5 * torch.randn(2 * I, generator=g)).to(dtype).cuda()
    wp, bp = P.pack_geglu_weight(w, b)
    y = P.gemm(x, wp, bp, mode="geglu")
    assert y.shape == (M, I)
    h = (x.float() @ w.float().T + b.float()).to(dtype).float()          # torch materialises the projection in the storage dtype
    ref = h[:, :I] * F.gelu(h[:, I:])
    assert (y.float() - ref).abs().max().item() <= 2 * _tol(dtype) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,S,K,N", [(3, 256, 320, 320), (2, 1024, 640, 640), (5, 64, 1280, 1280), (2, 128, 1024, 320), (1, 4096, 320, 320), (7, 64, 320, 640)])
def test_gemm_transposed_output(P, dtype, B, S, K, N):
    """the attention kernel's V^T operand: [B, S, K] x [N, K]^T -> [B, N, S]"""
    g = torch.Generator().manual_seed(B + S + N)
    x = torch.randn(B, S, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    y = P.gemm(x, w, b, mode="trans", tokens=S)
    assert y.shape == (B, N, S)
    ref = (x.float() @ w.float().T + b.float()).transpose(1, 2)
    assert (y.float() - ref).abs().max().item() <= _tol(dtype) * ref.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,K,N", [(2, 16, 16, 320, 320), (3, 8, 8, 1280, 1280), (1, 64, 64, 320, 320), (2, 6, 10, 640, 640)])
def test_gemm_tokens_to_pf_in_place_residual(P, dtype, B, H, W, K, N):
    """Transformer2DModel's `proj_out(tokens) + residual` written straight into the padded-flat NHWC tensor"""
    g = torch.Generator().manual_seed(B + H + N)
    tok = torch.randn(B, H * W, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    base = torch.randn(B, N, H, W, generator=g).to(dtype).cuda()
    x = P.PF.from_nchw(base)
    guard0 = x.buf[:x.G].clone()
    P.gemm(tok, w, b, resid=x.rows, mode="tok2pf", tokens=H * W, width=W, out=x.rows)
    ref = (tok.float() @ w.float().T + b.float()).view(B, H, W, N).permute(0, 3, 1, 2) + base.float()
    assert (x.to_nchw().float() - ref).abs().max().item() <= 2 * _tol(dtype) * ref.abs().max().item()
    grid = x.grid
    assert grid[:, 0].abs().max() == 0 and grid[:, -1].abs().max() == 0 and grid[:, :, 0].abs().max() == 0 and grid[:, :, -1].abs().max() == 0
    assert torch.equal(x.buf[:x.G], guard0)


def test_gemm_rejects_bad_operands(P):
    x = torch.randn(64, 320).half().cuda()
    w = torch.randn(320, 320).half().cuda()
    with pytest.raises(ValueError):
        P.gemm(x, w.float(), None)                       # dtype mismatch must not be read as fp16 bytes
    with pytest.raises(ValueError):
        P.gemm(x, w, torch.zeros(100).half().cuda())     # short bias
    with pytest.raises(RuntimeError):
        P.gemm(x, w.cpu(), None)
    with pytest.raises(Exception):
        P.gemm(torch.randn(64, 100).half().cuda(), torch.randn(320, 100).half().cuda(), None)    # K % 64
    with pytest.raises(Exception):
        P.gemm(x, torch.randn(100, 320).half().cuda(), None)                                     # N % 160
    with pytest.raises(ValueError):
        P.gemm(x, w, torch.zeros(328).half().cuda()[4:324])                                      # bias 8 bytes off a 16-byte boundary (LDS-DMA pieces)


# ---- N that is not a multiple of the 160-column tile (VAE: 128 / 256 / 512 channels), strided operands, row softmax, single-head attention
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(300, 128, 8), (1000, 512, 512), (4096, 512, 1024), (777, 256, 136), (64, 64, 168)])
def test_gemm_partial_last_column_tile(P, dtype, M, K, N):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dtype).cuda()
    b = torch.randn(N, generator=g).to(dtype).cuda()
    r = torch.randn(M, N, generator=g).to(dtype).cuda()
    guard = torch.full((M, N + 24), 7.0, dtype=dtype, device="cuda")
    y = P.gemm(x, w, b, resid=r)
    ref = x.float() @ w.float().t() + b.float() + r.float()
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    # a strided destination: nothing is written past column N
    P.gemm_strided(x, w, guard[:, :N], b)
    assert (guard[:, N:] == 7.0).all()
    assert (guard[:, :N].float() - (ref - r.float())).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    yt = P.gemm(x.view(1, M, K) if M % 8 == 0 else x[: M // 8 * 8].reshape(1, -1, K).contiguous(), w, b, mode="trans", tokens=M // 8 * 8)
    reft = (x[: M // 8 * 8].float() @ w.float().t() + b.float()).t()
    assert yt.shape == (1, N, M // 8 * 8) and (yt[0].float() - reft).abs().max().item() <= tol * max(1.0, reft.abs().max().item())


def test_gemm_strided_column_slices(P):
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(512, 1024, generator=g).half().cuda()            # [q | k] of a fused projection
    q, k = wide[:, :512], wide[:, 512:]
    out = torch.empty(512, 512, dtype=torch.float16, device="cuda")
    P.gemm_strided(q, k, out)
    ref = q.float() @ k.float().t()
    assert (out.float() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()
    with pytest.raises(Exception):
        P.gemm_strided(wide[:, 1:513], k, out)                        # misaligned slice


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(7, 8), (300, 4096), (64, 9216), (5, 1000)])
def test_softmax_rows(P, dtype, rows, cols):
    g = torch.Generator().manual_seed(cols)
    x = (4.0 * torch.randn(rows, cols + 8, generator=g)).to(dtype).cuda()
    keep = x.clone()
    v = x[:, :cols]
    P.softmax_rows_(v, 0.37)
    ref = torch.softmax(0.37 * keep[:, :cols].float(), dim=-1)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    assert (v.float() - ref).abs().max().item() <= tol * ref.max().item() + 1e-6
    assert torch.equal(x[:, cols:], keep[:, cols:])                      # the row stride is respected: nothing past `cols` is touched
    assert (v.float().sum(-1) - 1).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,S,d", [(2, 256, 512), (1, 1024, 128), (3, 64, 64), (2, 48, 128)])
def test_attention_single_head_vs_fp32(P, B, S, d):
    g = torch.Generator().manual_seed(S)
    qk = torch.randn(B, S, 2 * d, generator=g).half().cuda()
    v = torch.randn(B, S, d, generator=g).half().cuda()
    o = P.attention_single_head(qk[..., :d], qk[..., d:], v.transpose(1, 2).contiguous())
    qf, kf, vf = qk[..., :d].float(), qk[..., d:].float(), v.float()
    ref = torch.softmax(qf @ kf.transpose(1, 2) * d ** -0.5, dim=-1) @ vf
    assert (o.float() - ref).abs().max().item() <= 6e-3 * max(1.0, ref.abs().max().item())
