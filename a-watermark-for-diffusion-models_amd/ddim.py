"""DDIM (eta = 0) sampling and inversion loops -- rows G1 / X2 of SURVEY.md section 8a.

Reference: extract.py:49-69 runs a stock diffusers 0.26.0 `StableDiffusionPipeline` whose scheduler is
`DDIMInverseScheduler` (prompt "", guidance_scale 1, fp16, output_type 'latent'); the older closed form survives as
bytecode in __pycache__/inverse_stable_diffusion_gs.cpython-38.pyc (`backward_ddim`, `backward_diffusion`).  The step
arithmetic and the loop structure below (timestep order, prev_timestep, final_alpha_cumprod, the alpha swap of the
inversion, classifier-free guidance) are PINNED against vectors obtained by executing that bytecode
(tests/golden/make_golden_ddim.py, tests/test_ddim_bytecode_golden.py, tests/test_gpu_ddim.py); the timestep list and the
alpha table themselves are the published SD scheduler config (scaled-linear betas 0.00085..0.012, 1000 train steps,
'leading' spacing, steps_offset 1, set_alpha_to_one False) -- diffusers itself is not available here.

MI355X design: the per-step scalars (a_t, b_t) are computed once on the host in fp64; each step is ONE UNet evaluation
(the hand-written MFMA kernels of libgswm) plus ONE fused HIP kernel for the whole scheduler step
(x' = a x + b eps, with classifier-free guidance folded in when sampling: libgswm `gsw_ddim_step[_cfg]`), and the last
inversion step is fused with the Gaussian-CDF quantiser and the majority vote (`gsw_ddim_step_extract`) so the inverted
latent never makes an extra trip through HBM -- and never visits the host (the reference's `.cpu()`, extract.py:70).
Timesteps live on the device; there is no host synchronisation inside a loop, so a loop can be captured in a HIP graph.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Tuple

import numpy as np
import torch

from . import codec


def sd_alphas_cumprod(num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> np.ndarray:
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def step_coefficients(alpha_from: float, alpha_to: float, prediction_type: str = "epsilon") -> Tuple[float, float]:
    """(a, b) with x_to = a * x_from + b * model_out for the deterministic DDIM move alpha_from -> alpha_to."""
    af, at = float(alpha_from), float(alpha_to)
    if prediction_type == "epsilon":
        return (at / af) ** 0.5, (1 - at) ** 0.5 - (at * (1 - af) / af) ** 0.5
    if prediction_type == "v_prediction":
        return (at * af) ** 0.5 + ((1 - at) * (1 - af)) ** 0.5, ((1 - at) * af) ** 0.5 - (at * (1 - af)) ** 0.5
    raise ValueError(prediction_type)


@dataclass
class DDIMSchedule:
    num_inference_steps: int = 50
    num_train_timesteps: int = 1000
    steps_offset: int = 1
    prediction_type: str = "epsilon"
    set_alpha_to_one: bool = False
    beta_start: float = 0.00085
    beta_end: float = 0.012

    def __post_init__(self):
        self.alphas_cumprod = sd_alphas_cumprod(self.num_train_timesteps, self.beta_start, self.beta_end)
        self.final_alpha = 1.0 if self.set_alpha_to_one else float(self.alphas_cumprod[0])
        if not 1 <= self.num_inference_steps <= self.num_train_timesteps:      # (before the division: 0 steps would be a ZeroDivisionError, more steps than training timesteps a ratio of 0)
            raise ValueError(f"DDIMSchedule: num_inference_steps must be in 1..{self.num_train_timesteps} (got {self.num_inference_steps})")
        self.ratio = self.num_train_timesteps // self.num_inference_steps
        ts = (np.arange(self.num_inference_steps) * self.ratio).round().astype(np.int64) + self.steps_offset
        self.timesteps_desc = ts[::-1].copy()      # sampling order: 981, 961, ..., 1 for 50 steps
        if int(ts.min()) < 0 or int(ts.max()) >= self.num_train_timesteps:
            # the eps model tabulates its time embedding over [0, num_train_timesteps) (unet.TEMB_TABLE): a timestep outside it has no row
            raise ValueError(f"DDIMSchedule: timesteps {int(ts.min())}..{int(ts.max())} leave [0, {self.num_train_timesteps}) "
                             f"({self.num_inference_steps} steps, steps_offset {self.steps_offset})")

    def _alpha(self, t: int) -> float:
        return float(self.alphas_cumprod[t]) if t >= 0 else self.final_alpha

    def sampling(self) -> List[Tuple[int, float, float]]:
        """[(t, a, b)]: the model is evaluated at t, the sample moves alpha[t] -> alpha[t - ratio]."""
        return [(int(t), *step_coefficients(self._alpha(int(t)), self._alpha(int(t) - self.ratio), self.prediction_type))
                for t in self.timesteps_desc]

    def inversion(self) -> List[Tuple[int, float, float]]:
        """[(t, a, b)] ascending: the model is evaluated at t, the sample moves alpha[t - ratio] -> alpha[t]
        (DDIMInverseScheduler.step; == backward_diffusion(reverse_process=True) of the recovered bytecode)."""
        return [(int(t), *step_coefficients(self._alpha(int(t) - self.ratio), self._alpha(int(t)), self.prediction_type))
                for t in self.timesteps_desc[::-1]]


EpsModel = Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]


def _t_tensors(ts, device):
    # one small device tensor per step, created before the loop: no host->device traffic inside it
    return [torch.full((), t, dtype=torch.int64, device=device) for t in ts]


def _cfg_contexts(ctx_uncond: torch.Tensor, ctx_text: torch.Tensor) -> torch.Tensor:
    """(uncond | text) contexts of a guidance batch as ONE tensor -- the SAME object for the same pair of inputs: the eps model keeps its per-context caches ON the context
    tensor (padded copy, cross-attention K / V^T, the one-launch cross-attention's fragment streams: unet.py, xattn.py), so a second sampling loop over the same prompts finds
    them instead of rebuilding them.  The entry holds a reference to ctx_uncond (its storage cannot be recycled under the key) and follows both tensors' version counters."""
    key = (id(ctx_uncond), ctx_uncond._version, tuple(ctx_uncond.shape), tuple(ctx_uncond.stride()), ctx_text._version, tuple(ctx_text.shape))
    c = getattr(ctx_text, "_gsw_cfg_cat", None)
    if c is None or c[0] != key:
        c = (key, torch.cat([ctx_uncond, ctx_text], dim=0), ctx_uncond)
        ctx_text._gsw_cfg_cat = c
    return c[1]


@torch.no_grad()
def ddim_sample(eps_model: EpsModel, z_T: torch.Tensor, ctx_text: torch.Tensor, schedule: DDIMSchedule, *,
                ctx_uncond: Optional[torch.Tensor] = None, guidance_scale: float = 7.5) -> torch.Tensor:
    """G1: txt2img latent loop starting from the watermarked Z_s_T.  guidance_scale != 1 runs the usual 2B-row UNet batch
    (uncond | text) and folds `uncond + g (text - uncond)` into the scheduler-step kernel."""
    steps = schedule.sampling()
    tt = _t_tensors([s[0] for s in steps], z_T.device)
    x = z_T.clone()
    use_cfg = guidance_scale != 1.0 and ctx_uncond is not None
    ctx2 = _cfg_contexts(ctx_uncond, ctx_text) if use_cfg else None
    B = x.shape[0]
    # an eps model that knows the two halves of the guidance batch are the same latents (unet.UNet2DCondition: cfg_dup) computes what does not depend on
    # the context once; any other callable gets the reference's doubled batch (`torch.cat([latents] * 2)`, modified_stable_diffusion_gs.pyc)
    dup = bool(getattr(eps_model, "supports_cfg_dup", False))
    for (t, a, b), t_dev in zip(steps, tt):
        if use_cfg:
            out = eps_model(x, t_dev, ctx2, cfg_dup=True) if dup else eps_model(torch.cat([x, x], dim=0), t_dev, ctx2)
            codec.ddim_step_cfg(x, out[:B], out[B:], a, b, guidance_scale, out=x)
        else:
            codec.ddim_step(x, eps_model(x, t_dev, ctx_text), a, b, out=x)
    return x


@torch.no_grad()
def ddim_invert(eps_model: EpsModel, x0: torch.Tensor, ctx: torch.Tensor, schedule: DDIMSchedule) -> torch.Tensor:
    """X2: DDIM inversion x_0 -> x_T (prompt "", guidance 1: one UNet evaluation per step, extract.py:66-69)."""
    steps = schedule.inversion()
    tt = _t_tensors([s[0] for s in steps], x0.device)
    x = x0.clone()
    for (t, a, b), t_dev in zip(steps, tt):
        codec.ddim_step(x, eps_model(x, t_dev, ctx), a, b, out=x)
    return x


@torch.no_grad()
def ddim_invert_extract(eps_model: EpsModel, x0: torch.Tensor, ctx: torch.Tensor, schedule: DDIMSchedule, key: bytes, nonce: bytes,
                        message_length: int, *, return_latents: bool = False, return_counts: bool = False):
    """X2 + X3-X5 in one pass: the last scheduler step, the Gaussian-CDF quantiser and the vote are one kernel."""
    steps = schedule.inversion()
    tt = _t_tensors([s[0] for s in steps], x0.device)
    x = x0.clone()
    for (t, a, b), t_dev in zip(steps[:-1], tt[:-1]):
        codec.ddim_step(x, eps_model(x, t_dev, ctx), a, b, out=x)
    t, a, b = steps[-1]
    eps = eps_model(x, tt[-1], ctx)
    z_out = torch.empty_like(x) if return_latents else None
    res = codec.ddim_step_extract(x, eps, a, b, key, nonce, message_length, z_out=z_out, return_counts=return_counts)
    return (*res, z_out) if return_latents else res


# =====================================================================================================================
# `--scheduler DPMs` of the reference (extract.py:49-50): diffusers' DPMSolverMultistepInverseScheduler with its defaults on the
# SD scheduler config -- DPM-Solver++ (data prediction), solver_order 2, midpoint, 'linspace' timestep spacing,
# lower_order_final (only effective below 15 steps), no Karras sigmas.  Restated from the published algorithm
# (Lu et al., "DPM-Solver++", multistep 2M) in diffusers' sigma parametrisation; parity unpinned: this scheduler exists only inside diffusers (the DDIM loop, by contrast, is pinned against the reference bytecode).
# Every update is linear in (x, eps, previous x0 prediction), so a step is 2-3 launches of the fused scheduler-step kernel.
# =====================================================================================================================
@dataclass
class DPMSolverInverseSchedule:
    num_inference_steps: int = 50
    num_train_timesteps: int = 1000
    prediction_type: str = "epsilon"
    solver_order: int = 2
    lower_order_final: bool = True

    def __post_init__(self):
        ac = sd_alphas_cumprod(self.num_train_timesteps)
        S = self.num_inference_steps
        noisiest = self.num_train_timesteps - 1
        ts = np.linspace(0, noisiest, S + 1).round()[:-1].astype(np.int64)            # ascending: 0 ... < 999
        all_sigmas = ((1 - ac) / ac) ** 0.5
        sig = np.interp(ts, np.arange(len(all_sigmas)), all_sigmas)
        self.timesteps = ts
        self.sigmas = np.concatenate([sig, [all_sigmas[noisiest]]])                   # sigma_{S} = the noisiest level

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1.0 / (sigma * sigma + 1.0) ** 0.5
        return alpha, sigma * alpha

    def steps(self):
        """[(t, (P, Q), (A, B, C))]: m0 = P x + Q eps (x0 prediction); x' = A x + B m0 + C m1 (m1 = previous x0 prediction)."""
        out = []
        S = self.num_inference_steps
        for i, t in enumerate(self.timesteps):
            a_s0, s_s0 = self._alpha_sigma(self.sigmas[i])
            a_t, s_t = self._alpha_sigma(self.sigmas[i + 1])
            if self.prediction_type == "epsilon":
                P, Q = 1.0 / a_s0, -s_s0 / a_s0
            elif self.prediction_type == "v_prediction":
                P, Q = a_s0, -s_s0
            else:
                raise ValueError(self.prediction_type)
            lam_t, lam_s0 = np.log(a_t) - np.log(s_t), np.log(a_s0) - np.log(s_s0)
            h = lam_t - lam_s0
            A = s_t / s_s0
            k = -a_t * (np.exp(-h) - 1.0)
            first_order = self.solver_order == 1 or i == 0 or (self.lower_order_final and i == S - 1 and S < 15)
            if first_order:
                B, C = k, 0.0
            else:
                a_s1, s_s1 = self._alpha_sigma(self.sigmas[i - 1])
                lam_s1 = np.log(a_s1) - np.log(s_s1)
                r0 = (lam_s0 - lam_s1) / h
                B, C = k * (1.0 + 0.5 / r0), -k * 0.5 / r0
            out.append((int(t), (float(P), float(Q)), (float(A), float(B), float(C))))
        return out


@torch.no_grad()
def dpms_invert(eps_model: EpsModel, x0: torch.Tensor, ctx: torch.Tensor, schedule: DPMSolverInverseSchedule) -> torch.Tensor:
    """X2 with `--scheduler DPMs`: DPM-Solver++(2M) inversion x_0 -> x_T."""
    steps = schedule.steps()
    tt = _t_tensors([s[0] for s in steps], x0.device)
    x = x0.clone()
    m_prev = None
    for (t, (P, Q), (A, B, C)), t_dev in zip(steps, tt):
        m0 = codec.ddim_step(x, eps_model(x, t_dev, ctx), P, Q)          # x0 prediction of this step
        codec.ddim_step(x, m0, A, B, out=x)
        if C != 0.0:
            codec.ddim_step(x, m_prev, 1.0, C, out=x)
        m_prev = m0
    return x
