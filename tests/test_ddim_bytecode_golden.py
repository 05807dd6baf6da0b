"""CPU: the DDIM arithmetic of the product (ddim.py) and of the oracle (gs_oracle.py) against vectors produced by EXECUTING the reference's
own shipped bytecode (inverse_stable_diffusion_gs.cpython-38.pyc: backward_ddim, forward_ddim, backward_diffusion) -- see
tests/golden/make_golden_ddim.py for how the bytecode was run and which inputs (timestep list, alpha table, eps function) are stubs."""
import json
import os

import numpy as np
import pytest

import gs_oracle as O
import gswm_amd
from gswm_amd import ddim

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ddim_bytecode.json")))


def eps_fn(x, t):
    return 0.3 * np.tanh(x) + 0.05 * np.sin(3.0 * x + 0.01 * t)


def eps_text_fn(x, t):
    return eps_fn(x, t) + 0.1 * np.cos(2.0 * x - 0.003 * t)


def test_fixture_provenance():
    assert "cpython-38.pyc" in GOLD["source"] and GOLD["reencoded_is_not"] == 3
    assert len(GOLD["step"]) == 8 and len(GOLD["loops"]) == 12
    ac = O.sd_alphas_cumprod()
    for k, v in GOLD["alphas_cumprod_probe"].items():           # the alpha table handed to the bytecode is the one both restatements compute
        assert ac[int(k)] == v and ddim.sd_alphas_cumprod()[int(k)] == v


@pytest.mark.parametrize("i", range(8))
def test_single_step_closed_form(i):
    g = GOLD["step"][i]
    x, e, want = np.array(g["x_t"]), np.array(g["eps"]), np.array(g["out"])
    np.testing.assert_allclose(O.backward_ddim(x, g["alpha_t"], g["alpha_tm1"], e), want, rtol=1e-14, atol=1e-15)
    for coeff in (O.ddim_coefficients, ddim.step_coefficients):   # the (a, b) form the kernels consume
        a, b = coeff(g["alpha_t"], g["alpha_tm1"])
        np.testing.assert_allclose(a * x + b * e, want, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("i", range(12))
def test_loop_structure_and_result(i):
    g = GOLD["loops"][i]
    S, rev, gs = g["steps"], g["reverse_process"], g["guidance_scale"]
    sched = ddim.DDIMSchedule(num_inference_steps=S)
    mine = sched.inversion() if rev else sched.sampling()
    ref = O.ddim_schedule(S, inverse=rev)
    # timestep order the model is evaluated at, per step
    assert [t for t, _, _ in mine] == g["model_t"] == [t for t, _, _ in ref]
    # per-step alphas: the pair the bytecode handed to backward_ddim gives the same (a, b) as both schedules
    for (t, a, b), (_, ra, rb), af, at in zip(mine, ref, g["alpha_t"], g["alpha_tm1"]):
        wa, wb = ddim.step_coefficients(af, at)
        assert abs(a - wa) <= 1e-15 * abs(wa) and abs(b - wb) <= 1e-13 * max(1e-3, abs(wb))
        assert abs(ra - wa) <= 1e-15 * abs(wa) and abs(rb - wb) <= 1e-13 * max(1e-3, abs(wb))
    # the whole loop in float64 with the same analytic eps model
    x = np.array(g["x_in"])
    for t, a, b in mine:
        e = eps_fn(x, t)
        if gs > 1.0:
            e = e + gs * (eps_text_fn(x, t) - e)
        x = a * x + b * e
    np.testing.assert_allclose(x, np.array(g["x_out"]), rtol=0, atol=1e-10)
