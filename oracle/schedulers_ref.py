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


# ---------------------------------------------------------------------------------------------------
# The other five names of the reference's registry (`models/stable_diffusion.py:199-227`).  PARITY UNPINNED
# like the three above: restated from the published algorithms (Karras et al. 2022 ancestral sampler and
# rho = 7 ladder; Lu et al. DPM-Solver++ 2M and its SDE variant; Liu et al. PNDM / PLMS; Zhao et al. UniPC
# with B(h) = e^{-h} - 1) in the form diffusers 0.27.2 gives them.  Noise for the stochastic ones is handed in.
# ---------------------------------------------------------------------------------------------------
class EulerAncestralRef(EulerRef):
    def step(self, eps, t, x, noise):
        s, s_to = self.sigmas[self.i], self.sigmas[self.i + 1]
        x0 = x - s * eps
        s_up = np.sqrt(s_to ** 2 * (s ** 2 - s_to ** 2) / s ** 2)
        s_down = np.sqrt(s_to ** 2 - s_up ** 2)
        d = (x - x0) / s
        self.i += 1
        return x + d * (s_down - s) + noise * s_up


def karras_ladder(sig_all, n, rho=7.0):
    lo, hi = sig_all[0] ** (1.0 / rho), sig_all[-1] ** (1.0 / rho)
    return np.array([(hi + k / (n - 1) * (lo - hi)) ** rho for k in range(n)]) if n > 1 else np.array([sig_all[-1]])


def sigma_to_timestep(sigma, sig_all):
    """Piecewise-linear inverse of log sigma(t) over the integer training timesteps."""
    ls, la = np.log(max(sigma, 1e-10)), np.log(sig_all)
    lo = 0
    for k in range(len(la)):
        if ls >= la[k]:
            lo = k
    lo = min(lo, len(la) - 2)
    w = min(max((la[lo] - ls) / (la[lo] - la[lo + 1]), 0.0), 1.0)
    return (1 - w) * lo + w * (lo + 1)


class DPMpp2MKarrasRef(DPMpp2MRef):
    def set_timesteps(self, n):
        self.n = n
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = karras_ladder(sig_all, n)
        self.timesteps = np.array([round(sigma_to_timestep(s, sig_all)) for s in sig], dtype=np.int64)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.i = 0
        self.m_prev = None
        return self.timesteps


class DPMpp2MSDERef(DPMpp2MRef):
    """sde-dpmsolver++ (2M, midpoint) on the ordinary (non-Karras) schedule: see the product class's note on the
    reference's misspelt keyword."""

    def step(self, eps, t, x, noise):
        i = self.i
        a0, sg0 = self._alpha_sigma(self.sigmas[i])
        a_t, sg_t = self._alpha_sigma(self.sigmas[i + 1])
        m0 = (x - sg0 * eps) / a0
        lam_0 = np.log(a0) - np.log(sg0)
        lam_t = np.log(a_t) - np.log(sg_t) if sg_t > 0 else np.inf
        h = lam_t - lam_0
        out = (sg_t / sg0 * np.exp(-h)) * x + a_t * (1 - np.exp(-2 * h)) * m0 + sg_t * np.sqrt(1 - np.exp(-2 * h)) * noise
        if not (i == self.n - 1 or self.m_prev is None):
            a1, sg1 = self._alpha_sigma(self.sigmas[i - 1])
            r0 = (lam_0 - (np.log(a1) - np.log(sg1))) / h
            d1 = (m0 - self.m_prev) / r0
            out = out + 0.5 * a_t * (1 - np.exp(-2 * h)) * d1
        self.m_prev = m0
        self.i += 1
        return out


class PNDMRef:
    """PLMS: pseudo linear multistep on epsilon, warm-up by one extra evaluation at the second timestep."""
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, T=1000, steps_offset=1):
        self.T, self.ac, self.steps_offset = T, alphas_cumprod(T), steps_offset
        self.final_alpha = self.ac[0]

    def set_timesteps(self, n):
        self.n = n
        base = leading_timesteps(n, self.T, self.steps_offset)            # descending
        self.timesteps = np.concatenate([base[:1], base[1:2], base[1:]])   # t0, t1, t1, t2, ...
        self.ets, self.counter, self.cur = [], 0, None
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def _transfer(self, x, t, tp, e):
        a_t = self.ac[t]
        a_p = self.ac[tp] if tp >= 0 else self.final_alpha
        return np.sqrt(a_p / a_t) * x - (a_p - a_t) * e / (a_t * np.sqrt(1 - a_p) + np.sqrt(a_t * (1 - a_t) * a_p))

    def step(self, eps, t, x):
        t = int(t)
        r = self.T // self.n
        tp = t - r
        if self.counter != 1:
            self.ets = (self.ets + [eps])[-4:]
        else:
            tp, t = t, t + r
        k = len(self.ets)
        if k == 1 and self.counter == 0:
            e, self.cur = eps, x
        elif k == 1 and self.counter == 1:
            e, x, self.cur = (eps + self.ets[-1]) / 2, self.cur, None
        elif k == 2:
            e = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif k == 3:
            e = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            e = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        self.counter += 1
        return self._transfer(x, t, tp, e)


class UniPCRef:
    """UniPC-2 with B(h) = exp(-h) - 1 ("bh2"), data prediction, corrector on every step after the first,
    order lowered to 1 on the first and the last step."""
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, T=1000, steps_offset=1, spacing="leading"):
        self.T, self.ac, self.steps_offset, self.spacing = T, alphas_cumprod(T), steps_offset, spacing

    def set_timesteps(self, n):
        self.n = n
        if self.spacing == "linspace":
            ts = np.linspace(0, self.T - 1, n + 1).round()[::-1][:-1].astype(np.int64)
        else:
            ts = (np.arange(0, n + 1) * (self.T // (n + 1))).round()[::-1][:-1].astype(np.int64) + self.steps_offset
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        self.sigmas = np.concatenate([np.interp(ts, np.arange(len(sig_all)), sig_all), [sig_all[0]]])
        self.timesteps = ts
        self.i, self.hist, self.last_x, self.prev_order = 0, [], None, 1
        return ts

    def scale_model_input(self, x, t):
        return x

    @staticmethod
    def _lam(s):
        a = 1.0 / np.sqrt(s * s + 1.0)
        return a, s * a, np.log(a) - np.log(s * a)

    @staticmethod
    def _coeffs(h, order):
        hh = -h
        phi1 = np.expm1(hh)
        B = np.expm1(hh)
        phik, fact, b = phi1 / hh - 1, 1.0, []
        for k in range(1, order + 1):
            b.append(phik * fact / B)
            fact *= k + 1
            phik = phik / hh - 1.0 / fact
        return phi1, B, np.array(b)

    def step(self, eps, t, x):
        i = self.i
        a_i, s_i, lam_i = self._lam(self.sigmas[i])
        x0 = (x - s_i * eps) / a_i
        if i > 0:                                            # UniC: correct x with the new data prediction
            a_p, s_p, lam_p = self._lam(self.sigmas[i - 1])
            h = lam_i - lam_p
            phi1, B, b = self._coeffs(h, self.prev_order)
            m0 = self.hist[-1]
            base = s_i / s_p * self.last_x - a_i * phi1 * m0
            if self.prev_order == 1:
                x = base - a_i * B * 0.5 * (x0 - m0)
            else:
                rk = (self._lam(self.sigmas[i - 2])[2] - lam_p) / h
                rho = np.linalg.solve(np.array([[1.0, 1.0], [rk, 1.0]]), b)
                x = base - a_i * B * (rho[0] * (self.hist[-2] - m0) / rk + rho[1] * (x0 - m0))
        self.hist = (self.hist + [x0])[-2:]
        order = min(2, self.n - i, len(self.hist))
        self.prev_order, self.last_x = order, x
        a_t, s_t, lam_t = self._lam(self.sigmas[i + 1])        # UniP: predict the next sample
        h = lam_t - lam_i
        phi1, B, _ = self._coeffs(h, order)
        out = s_t / s_i * x - a_t * phi1 * x0
        if order == 2:
            rk = (self._lam(self.sigmas[i - 1])[2] - lam_i) / h
            out = out - a_t * B * 0.5 * (self.hist[-2] - x0) / rk
        self.i += 1
        return out
