"""ORACLE -- test infrastructure only.  numpy float64 restatement of the scheduler arithmetic.

PARITY UNPINNED vs diffusers==0.27.2 (absent; see oracle/unet_ref.py).  Follows the published
`DDIMScheduler` / `DPMSolverMultistepScheduler` / `EulerDiscreteScheduler` algorithms with the
constants the reference restates at `/root/reference/scripts/convert_from_A1111.py:947-959`
(`scaled_linear`, beta 0.00085..0.012, T=1000, steps_offset=1, set_alpha_to_one=False,
clip_sample=False) and the registry at `/root/reference/models/stable_diffusion.py:199-227`.
Call sites: `/root/reference/pipelines/sd_unified_pipeline.py:203-207` (set_timesteps),
`:472` (scale_model_input), `:489` (step), `:841` (add_noise).

Written independently of stablediffusion_amd/schedulers.py (torch, product host code) so the two
can be checked against each other.
"""
from __future__ import annotations

import numpy as np


def alphas_cumprod(T=1000, beta_start=0.00085, beta_end=0.012):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


def leading_timesteps(n, T=1000, steps_offset=1):
    ratio = T // n
    return (np.arange(0, n) * ratio).round()[::-1].astype(np.int64) + steps_offset


class DDIMRef:
    """eta = 0 DDIM, epsilon prediction."""
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, T=1000, steps_offset=1):
        self.T = T
        self.ac = alphas_cumprod(T)
        self.final_alpha = self.ac[0]          # set_alpha_to_one=False
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        self.n = n
        self.timesteps = leading_timesteps(n, self.T, self.steps_offset)
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x):
        prev = int(t) - self.T // self.n
        a_t = self.ac[int(t)]
        a_prev = self.ac[prev] if prev >= 0 else self.final_alpha
        x0 = (x - np.sqrt(1.0 - a_t) * eps) / np.sqrt(a_t)
        return np.sqrt(a_prev) * x0 + np.sqrt(1.0 - a_prev) * eps

    def add_noise(self, x0, noise, t):
        a = self.ac[int(t)]
        return np.sqrt(a) * x0 + np.sqrt(1.0 - a) * noise


class DPMpp2MRef:
    """DPM-Solver++(2M), midpoint, epsilon prediction, lower_order_final, final sigma = 0."""
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, T=1000, steps_offset=1, spacing="leading"):
        self.T = T
        self.ac = alphas_cumprod(T)
        self.steps_offset = steps_offset
        self.spacing = spacing

    def set_timesteps(self, n):
        self.n = n
        last = self.T  # lambda_min_clipped = -inf -> clipped_idx = 0
        if self.spacing == "linspace":
            ts = np.linspace(0, last - 1, n + 1).round()[::-1][:-1].astype(np.int64)
        else:
            ratio = last // (n + 1)
            ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].astype(np.int64) + self.steps_offset
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.timesteps = ts
        self.i = 0
        self.m_prev = None
        return ts

    def scale_model_input(self, x, t):
        return x

    def add_noise(self, x0, noise, t):
        """diffusers DPMSolverMultistepScheduler.add_noise: alpha_t x0 + sigma_t noise at the schedule
        position of `t`."""
        a, sg = self._alpha_sigma(self.sigmas[int(np.nonzero(self.timesteps == int(t))[0][0])])
        return a * x0 + sg * noise

    def start_at(self, t):
        """Loop entered mid-schedule (img2img strength < 1 / denoising_start): diffusers resolves the
        step index from the first timestep it is given (`_init_step_index`)."""
        self.i = int(np.nonzero(self.timesteps == int(t))[0][0])

    @staticmethod
    def _alpha_sigma(s):
        a = 1.0 / np.sqrt(s * s + 1.0)
        return a, s * a

    def step(self, eps, t, x):
        i = self.i
        s0 = self.sigmas[i]
        a0, sg0 = self._alpha_sigma(s0)
        m0 = (x - sg0 * eps) / a0
        s_t = self.sigmas[i + 1]
        a_t, sg_t = self._alpha_sigma(s_t)
        lam_t = np.log(a_t) - np.log(sg_t) if sg_t > 0 else np.inf
        lam_0 = np.log(a0) - np.log(sg0)
        h = lam_t - lam_0
        last = i == self.n - 1
        if i == 0 or last or self.m_prev is None:
            out = (sg_t / sg0) * x - a_t * (np.exp(-h) - 1.0) * m0
        else:
            s1 = self.sigmas[i - 1]
            a1, sg1 = self._alpha_sigma(s1)
            lam_1 = np.log(a1) - np.log(sg1)
            h0 = lam_0 - lam_1
            r0 = h0 / h
            d1 = (m0 - self.m_prev) / r0
            out = ((sg_t / sg0) * x - a_t * (np.exp(-h) - 1.0) * m0
                   - 0.5 * a_t * (np.exp(-h) - 1.0) * d1)
        self.m_prev = m0
        self.i += 1
        return out


class EulerRef:
    """EulerDiscreteScheduler (the reference's default, stable_diffusion.py:135-138), leading spacing."""
    order = 1

    def __init__(self, T=1000, steps_offset=1):
        self.T = T
        self.ac = alphas_cumprod(T)
        self.steps_offset = steps_offset

    def set_timesteps(self, n):
        self.n = n
        ts = leading_timesteps(n, self.T, self.steps_offset).astype(np.float64)
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.timesteps = ts
        # leading spacing: init_noise_sigma = sqrt(max_sigma^2 + 1)
        self.init_noise_sigma = float(np.sqrt(self.sigmas.max() ** 2 + 1.0))
        self.i = 0
        return ts

    def start_at(self, t):
        """See DPMpp2MRef.start_at."""
        self.i = int(np.nonzero(np.isclose(self.timesteps, float(t)))[0][0])

    def scale_model_input(self, x, t):
        s = self.sigmas[self.i]
        return x / np.sqrt(s * s + 1.0)

    def add_noise(self, x0, noise, t):
        """diffusers EulerDiscreteScheduler.add_noise: sigma-space, x0 + sigma(t) * noise, sigma looked up
        at the schedule position of `t` (index_for_timestep: first match)."""
        idx = int(np.nonzero(np.isclose(self.timesteps, float(t)))[0][0])
        return x0 + self.sigmas[idx] * noise

    def step(self, eps, t, x):
        s = self.sigmas[self.i]
        x0 = x - s * eps
        d = (x - x0) / s
        out = x + d * (self.sigmas[self.i + 1] - s)
        self.i += 1
        return out
