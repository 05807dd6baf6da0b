"""CPU: host-side DDIM schedule of the product (ddim.py) against the oracle's independent restatement (both follow the
closed form of inverse_stable_diffusion_gs.pyc; both are pinned against the bytecode itself in test_ddim_bytecode_golden.py)."""
import numpy as np
import pytest

import gs_oracle as O
import gswm_amd
from gswm_amd import ddim


@pytest.mark.parametrize("steps", [50, 30, 20, 10, 100])
@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_schedule_matches_oracle(steps, pred):
    s = ddim.DDIMSchedule(num_inference_steps=steps, prediction_type=pred)
    np.testing.assert_allclose(s.alphas_cumprod, O.sd_alphas_cumprod(), rtol=0, atol=0)
    for inverse, mine in ((False, s.sampling()), (True, s.inversion())):
        ref = O.ddim_schedule(steps, inverse=inverse, prediction_type=pred)
        assert [t for t, _, _ in mine] == [t for t, _, _ in ref]
        np.testing.assert_allclose([a for _, a, _ in mine], [a for _, a, _ in ref], rtol=1e-15)
        np.testing.assert_allclose([b for _, _, b in mine], [b for _, _, b in ref], rtol=1e-13, atol=1e-16)


def test_schedule_sd_constants():
    s = ddim.DDIMSchedule(50)
    assert list(s.timesteps_desc[:3]) == [981, 961, 941] and s.timesteps_desc[-1] == 1      # leading spacing, steps_offset 1
    assert abs(s.alphas_cumprod[0] - (1 - 0.00085)) < 1e-12 and abs(s.alphas_cumprod[-1] - 0.0046600) < 1e-6
    assert s.final_alpha == s.alphas_cumprod[0]                                               # set_alpha_to_one False
    # sampling and inversion are exact inverses of each other when the model output is held fixed
    x, e = np.random.RandomState(0).standard_normal((2, 100))
    xs = x.copy()
    for t, a, b in s.sampling():
        xs = a * xs + b * e
    for t, a, b in s.inversion():
        xs = a * xs + b * e
    np.testing.assert_allclose(xs, x, atol=1e-9)
    # closed form of the recovered bytecode
    t, a, b = s.sampling()[0]
    np.testing.assert_allclose(O.backward_ddim(x, s.alphas_cumprod[981], s.alphas_cumprod[961], e), a * x + b * e, atol=1e-12)


@pytest.mark.parametrize("steps", [50, 30, 10])
def test_dpms_inverse_coefficient_form_equals_stepwise_form(steps):
    """Product: per-step linear coefficients; oracle: diffusers-style step-by-step update.  Same eps function, float64."""
    sched = ddim.DPMSolverInverseSchedule(num_inference_steps=steps)
    assert sched.timesteps[0] == 0 and len(sched.timesteps) == steps and sched.timesteps[-1] < 999
    rng = np.random.RandomState(0)
    x0 = rng.standard_normal(500)
    eps_fn = lambda x, t: 0.3 * np.tanh(x) + 0.05 * np.sin(3 * x + 0.01 * t)
    ref = O.dpms_invert_reference(eps_fn, x0, steps)
    x, m_prev = x0.copy(), None
    for t, (P, Q), (A, B, C) in sched.steps():
        m0 = P * x + Q * eps_fn(x, t)
        x = A * x + B * m0 + (C * m_prev if C != 0.0 else 0.0)
        m_prev = m0
    np.testing.assert_allclose(x, ref, rtol=0, atol=1e-10)
    assert np.isfinite(ref).all()


@pytest.mark.parametrize("steps", [0, -3, 1001])
def test_schedule_refuses_a_step_count_outside_its_training_range(steps):
    """0 steps used to surface as a ZeroDivisionError from the ratio, more steps than training timesteps as a ratio of 0: both are ValueErrors of the schedule now"""
    with pytest.raises(ValueError, match="num_inference_steps"):
        ddim.DDIMSchedule(num_inference_steps=steps)
    assert ddim.DDIMSchedule(num_inference_steps=1000, steps_offset=0).ratio == 1 and ddim.DDIMSchedule(num_inference_steps=1, steps_offset=0).timesteps_desc.tolist() == [0]


def test_guidance_contexts_are_one_object_per_prompt_set():
    """ddim_sample concatenates (uncond | text) ONCE per pair of inputs: the eps model's per-context caches live on that tensor, so a second loop over the same prompts must
    get the same object -- and a new one when either input changes"""
    import torch
    from gswm_amd import pipeline
    u = torch.randn(1, 77, 8)
    t = torch.randn(3, 77, 8)
    p = pipeline.GaussianShadingPipeline.__new__(pipeline.GaussianShadingPipeline)
    p.ctx_uncond = u
    a, b = p._uncond(3), p._uncond(3)
    assert a is b and a.shape == (3, 77, 8) and p._uncond(2).shape == (2, 77, 8)
    c1, c2 = ddim._cfg_contexts(a, t), ddim._cfg_contexts(b, t)
    assert c1 is c2 and torch.equal(c1[:3], u.expand(3, -1, -1)) and torch.equal(c1[3:], t)
    t.add_(1.0)
    c3 = ddim._cfg_contexts(a, t)
    assert c3 is not c1 and torch.equal(c3[3:], t)
    p.ctx_uncond = torch.randn(1, 77, 8)
    assert p._uncond(3) is not a and ddim._cfg_contexts(p._uncond(3), t) is not c3
