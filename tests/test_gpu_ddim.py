"""GPU: the DDIM loops (G1 / X2) built on the fused HIP scheduler-step kernels, against a float64 numpy loop driven by the
oracle's schedule with the SAME eps function, and the end-to-end embed -> sample -> invert -> extract round trip on a small
UNet of the product's architecture."""
import numpy as np
import pytest
import torch

import gs_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    import gswm_amd
    from gswm_amd import codec, ddim, unet, pipeline
    import types
    return types.SimpleNamespace(codec=codec, ddim=ddim, unet=unet, pipeline=pipeline)


def analytic_eps(x, t, ctx):
    # smooth, cheap, deterministic stand-in for the UNet (fp32 math, result in x's dtype)
    xf = x.float()
    return (0.3 * torch.tanh(xf) + 0.05 * torch.sin(xf * 3.0 + t.float() * 0.01)).to(x.dtype)


def analytic_eps_np(x, t):
    return 0.3 * np.tanh(x) + 0.05 * np.sin(x * 3.0 + t * 0.01)


@pytest.mark.parametrize("steps", [10, 50])
def test_loops_vs_numpy_oracle_schedule(P, keys, steps):
    key, nonce = keys
    k = O.pad_message("lthero", 32)
    B = 3
    u = np.random.RandomState(1).uniform(0, 1, (B, 16384))
    zT = P.codec.embed_batch(key, nonce, k, B, (4, 64, 64), u=torch.from_numpy(u).cuda(), dtype=torch.float32)
    sched = P.ddim.DDIMSchedule(steps)
    ctx = torch.zeros(B, 1, 1, device="cuda")
    x0 = P.ddim.ddim_sample(analytic_eps, zT, ctx, sched, guidance_scale=1.0)
    zi = P.ddim.ddim_invert(analytic_eps, x0, ctx, sched)
    # float64 reference with the oracle's schedule
    x = zT.cpu().double().numpy()
    for t, a, b in O.ddim_schedule(steps, inverse=False):
        x = a * x + b * analytic_eps_np(x, t)
    np.testing.assert_allclose(x0.cpu().numpy(), x, rtol=0, atol=2e-4 * max(1.0, np.abs(x).max()))
    xi = x.copy()
    for t, a, b in O.ddim_schedule(steps, inverse=True):
        xi = a * xi + b * analytic_eps_np(xi, t)
    np.testing.assert_allclose(zi.cpu().numpy(), xi, rtol=0, atol=5e-4 * max(1.0, np.abs(xi).max()))
    # fused last step + vote == unfused inversion followed by extract, and equals the oracle's vote on the same latent
    bits, flags, zf = P.ddim.ddim_invert_extract(analytic_eps, x0, ctx, sched, key, nonce, 256, return_latents=True)
    assert torch.equal(zf, zi)
    b2, f2 = P.codec.extract_batch(zi, key, nonce, 256)
    assert torch.equal(bits, b2) and torch.equal(flags, f2)
    for b in range(B):
        assert P.codec.bits_to_str(bits[b].cpu().numpy()) == O.recover_bits(zi[b].cpu().numpy(), key, nonce, 256)


def test_cfg_sampling_equals_manual_guidance(P, keys):
    key, nonce = keys
    B = 2
    zT = P.codec.embed_batch(key, nonce, O.pad_message("x", 32), B, (4, 64, 64), seed=3, dtype=torch.float16)
    sched = P.ddim.DDIMSchedule(8)

    def eps_ctx(x, t, ctx):   # depends on the context so that uncond != text
        return (0.2 * torch.tanh(x.float()) * ctx.float().mean(dim=(1, 2))[:, None, None, None]).to(x.dtype)

    cu = torch.full((B, 2, 2), 0.5, device="cuda", dtype=torch.float16)
    ct = torch.full((B, 2, 2), 1.5, device="cuda", dtype=torch.float16)
    x0 = P.ddim.ddim_sample(eps_ctx, zT, ct, sched, ctx_uncond=cu, guidance_scale=7.5)
    x = zT.float()
    for t, a, b in sched.sampling():
        xh = x.half()
        eu, et = eps_ctx(xh, None, cu).float(), eps_ctx(xh, None, ct).float()
        x = (a * xh.float() + b * (eu + 7.5 * (et - eu))).half().float()
    assert (x0.float() - x).abs().max().item() <= 4e-3 * max(1.0, x.abs().max().item())


@pytest.mark.usefixtures("library_kernels_allowed")      # small / odd shapes off the hand-written path: strict mode (the default) would raise
def test_small_unet_roundtrip_is_lossless(P, keys):
    """embed -> 20-step CFG sampling -> 20-step inversion -> vote with a small UNet of the product architecture
    (synthetic weights): every bit of every image comes back."""
    key, nonce = keys
    k = O.pad_message("lthero", 32)
    torch.manual_seed(0)
    m = P.unet.UNet2DCondition(block_out_channels=(64, 128, 128, 128), cross_attention_dim=64, num_heads=(2, 4, 4, 4), head_dim=32)
    P.unet.synthetic_init_(m, 0)
    m = m.cuda().half().eval()
    g = torch.Generator().manual_seed(1)
    cu = torch.randn(1, 77, 64, generator=g).cuda().half()
    B = 4
    ct = torch.randn(B, 77, 64, generator=g).cuda().half()
    pipe = P.pipeline.GaussianShadingPipeline(m, key, nonce, k, num_inference_steps=20, ctx_uncond=cu)
    zT, x0, bits, flags = pipe.roundtrip(B, ct, seed=5, guidance_scale=7.5)
    assert zT.shape == (B, 4, 64, 64) and x0.shape == zT.shape and torch.isfinite(x0).all()
    assert int(flags.abs().sum()) == 0
    assert int(P.codec.bit_matches(bits, 256, k).min()) == 256
    zi = pipe.invert(x0)
    for b in range(B):
        assert P.codec.bits_to_str(bits[b].cpu().numpy()) == O.recover_bits(zi[b].cpu().numpy(), key, nonce, 256)
    assert ((zi >= 0) == (zT >= 0)).float().mean().item() > 0.9


def test_dpms_inversion_vs_numpy_oracle(P, keys):
    key, nonce = keys
    B = 2
    zT = P.codec.embed_batch(key, nonce, O.pad_message("x", 32), B, (4, 64, 64), seed=9, dtype=torch.float32)
    sched = P.ddim.DPMSolverInverseSchedule(20)
    ctx = torch.zeros(B, 1, 1, device="cuda")
    x0 = (zT * 0.2).contiguous()
    zi = P.ddim.dpms_invert(analytic_eps, x0, ctx, sched)
    ref = O.dpms_invert_reference(lambda x, t: analytic_eps_np(x, t), x0.cpu().double().numpy(), 20)
    np.testing.assert_allclose(zi.cpu().numpy(), ref, rtol=0, atol=1e-3 * max(1.0, np.abs(ref).max()))


# ---- the device loops against vectors produced by executing the reference's shipped bytecode (tests/golden/make_golden_ddim.py)
import json as _json
import os as _os

_BYTECODE = _json.load(open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "ddim_bytecode.json")))


def _bc_eps(x, t, ctx):
    xf = x.double()
    tf = t.double()
    e = 0.3 * torch.tanh(xf) + 0.05 * torch.sin(3.0 * xf + 0.01 * tf)
    if ctx is not None and ctx.shape[0] == x.shape[0] and bool(ctx.flatten()[-1] != 0):      # CFG batch: second half is the conditional branch
        h = x.shape[0] // 2
        e[h:] = e[h:] + 0.1 * torch.cos(2.0 * xf[h:] - 0.003 * tf)
    return e.to(x.dtype)


@pytest.mark.parametrize("i", range(12))
def test_device_loops_vs_reference_bytecode(P, i):
    """gsw_ddim_step / gsw_ddim_step_cfg inside ddim_sample / ddim_invert reproduce `backward_diffusion` of
    inverse_stable_diffusion_gs.pyc.  The device state is fp32, the bytecode ran in float64: after up to 50 steps (values reach |x| ~ 30 with
    guidance 7.5) the tolerance is 1e-5 absolute + 2e-6 relative, i.e. a few dozen fp32 ulps."""
    g = _BYTECODE["loops"][i]
    S, rev, gs = g["steps"], g["reverse_process"], g["guidance_scale"]
    sched = P.ddim.DDIMSchedule(S)
    x_in = torch.tensor(g["x_in"], dtype=torch.float32).cuda().contiguous()
    pad = (-x_in.numel()) % 8                                   # the step kernel works on 16-byte vectors
    xp = torch.cat([x_in.flatten(), torch.zeros(pad, device="cuda")]).view(2, -1).contiguous() if pad else x_in
    B = xp.shape[0]
    if rev:
        out = P.ddim.ddim_invert(_bc_eps, xp, torch.zeros(B, 1, device="cuda"), sched)
    elif gs > 1.0:
        out = P.ddim.ddim_sample(_bc_eps, xp, torch.ones(B, 1, device="cuda"), sched, ctx_uncond=torch.zeros(B, 1, device="cuda"), guidance_scale=gs)
    else:
        out = P.ddim.ddim_sample(_bc_eps, xp, torch.zeros(B, 1, device="cuda"), sched, guidance_scale=1.0)
    got = out.flatten()[: x_in.numel()].view_as(x_in).double().cpu().numpy()
    np.testing.assert_allclose(got, np.array(g["x_out"]), rtol=2e-6, atol=1e-5)
