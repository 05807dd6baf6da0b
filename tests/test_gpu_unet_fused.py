"""GPU: the fused elementwise kernels of the eps model against plain torch fp32 references of the same ops, and the whole
UNet with fusions on vs off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gswm_amd
    from gswm_amd import codec, unet
    import types
    return types.SimpleNamespace(codec=codec, unet=unet)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,C,H,W", [(3, 320, 64, 64), (2, 640, 32, 32), (2, 1280, 8, 8), (2, 960, 64, 64), (1, 2560, 16, 16), (2, 64, 4, 2), (2, 1920, 32, 32)])
@pytest.mark.parametrize("act,bias", [(True, True), (True, False), (False, False)])
def test_groupnorm_silu_vs_torch_fp32(G, dtype, B, C, H, W, act, bias):
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.3).to(dtype).cuda()
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    beta = (0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    pb = (0.5 * torch.randn(B, C, generator=g)).to(dtype).cuda() if bias else None
    y = G.codec.groupnorm_silu(x, gamma, beta, 32, 1e-5, act=act, pre_bias=pb)
    xf = x.float() + (pb.float()[:, :, None, None] if bias else 0)
    ref = F.group_norm(xf, 32, gamma.float(), beta.float(), 1e-5)
    ref = F.silu(ref) if act else ref
    tol = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]     # ~1 ulp of the storage dtype at |y| ~ 4
    assert y.shape == x.shape and y.dtype == dtype
    assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(2, 4096, 2560), (3, 77, 640), (1, 64, 10240), (5, 8, 16)])
def test_geglu_vs_torch_fp32(G, dtype, shape):
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 2).to(dtype).cuda()
    y = G.codec.geglu(x)
    h, gate = x.float().chunk(2, dim=-1)
    ref = h * F.gelu(gate)
    tol = {torch.float32: 2e-6, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    assert y.shape == (*shape[:-1], shape[-1] // 2)
    assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,C", [(4096 * 3, 320), (1024, 640), (257, 1280), (5, 64), (77, 1536), (33, 328), (9, 8), (130, 1288), (3, 648)])
@pytest.mark.parametrize("with_delta", [True, False])
def test_add_layernorm_vs_torch_fp32(G, dtype, rows, C, with_delta):
    g = torch.Generator().manual_seed(C + rows)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dtype).cuda()
    d = torch.randn(rows, C, generator=g).to(dtype).cuda() if with_delta else None
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    b = (0.2 * torch.randn(C, generator=g)).to(dtype).cuda()
    xn, y = G.codec.add_layernorm(x, d, w, b, 1e-5)
    xs = (x + d) if with_delta else x                           # the stored (rounded) sum is what gets normalised
    assert torch.equal(xn, xs)
    ref = F.layer_norm(xs.float(), (C,), w.float(), b.float(), 1e-5)
    tol = 4e-3 if dtype == torch.float16 else 3e-2
    assert (y.float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.usefixtures("library_kernels_allowed")      # small / odd shapes off the hand-written path: strict mode (the default) would raise
def test_unet_fused_equals_unfused(G):
    U = G.unet
    m = U.synthetic_init_(U.UNet2DCondition(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, num_heads=(2, 4, 4, 4), head_dim=32), 0)
    m = m.cuda().half().eval()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 4, 32, 32, generator=g).cuda().half()
    c = torch.randn(3, 77, 64, generator=g).cuda().half()
    t = torch.tensor([981, 500, 1]).cuda()
    with torch.no_grad():
        U.FUSED_KERNELS = True
        y1 = m(x, t, c)
        U.FUSED_KERNELS = False
        y0 = m(x, t, c)
        yref = m.float()(x.float(), t, c.float())            # plain torch fp32 end to end (FUSED_KERNELS still off)
        U.FUSED_KERNELS = True
    e1 = (y1.float() - yref).abs().max().item()
    e0 = (y0.float() - yref).abs().max().item()             # torch's own fp16 path: reported only
    assert e1 <= 1e-2 * max(1.0, yref.abs().max().item()), (e1, e0)      # absolute bound at the output's scale
