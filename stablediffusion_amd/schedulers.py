"""Host-side (PyTorch) schedulers with the diffusers call surface the reference pipeline uses.

`north_star` keeps the scheduler step on PyTorch-ROCm host code; in the reference these objects come
from diffusers (`/root/reference/models/stable_diffusion.py:199-227`) and are driven at
`/root/reference/pipelines/sd_unified_pipeline.py:203-207` (set_timesteps), `:472`
(scale_model_input), `:489` (step), `:502,:841` (add_noise), `:735` (.order / .timesteps),
`:785` (.init_noise_sigma), `:398` (.config.num_train_timesteps).
Constants follow `/root/reference/scripts/convert_from_A1111.py:947-959` (scaled_linear betas
0.00085..0.012, T=1000, steps_offset=1, set_alpha_to_one=False, clip_sample=False, epsilon).
Coefficients are evaluated in float64 on the host; tensors are updated in fp32 and cast back.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


class FusedPlan(SimpleNamespace):
    """One denoise step as the affine update the engine applies on the device (`sd_cfg_linear_step`):
        x0 = h_x x + h_eps eps;   x <- c_x x + c_eps eps + c_hist hist;   hist <- x0
    `in_scale` multiplies the UNet input (scale_model_input), `use_hist` says whether the scheduler
    carries an x0 history.  Coefficients are float64 host values."""


def _alphas_cumprod(T, beta_start, beta_end):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


class _Base:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1,
                 timestep_spacing="leading", **extra):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule="scaled_linear",
                                      steps_offset=steps_offset, timestep_spacing=timestep_spacing,
                                      prediction_type="epsilon", **extra)
        self.ac = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    @classmethod
    def from_config(cls, config, **kw):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        keep = {k: d[k] for k in ("num_train_timesteps", "beta_start", "beta_end", "steps_offset",
                                  "timestep_spacing") if k in d}
        keep.update(kw)
        return cls(**keep)

    def _leading(self, n, extra=0):
        T = self.config.num_train_timesteps
        ratio = T // (n + extra)
        ts = (np.arange(0, n + extra) * ratio).round()[::-1].copy().astype(np.int64)
        return ts + self.config.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _begin(self, timestep):
        """Position in the sigma schedule of the step being taken.  The first call after
        set_timesteps resolves it from the timestep (diffusers `_init_step_index` /
        `index_for_timestep`: with duplicates the second match, so that a loop started mid-schedule
        -- img2img with strength < 1, denoising_start; sd_unified_pipeline.py:722-761 slices
        `timesteps[t_start * order:]` -- does not skip a sigma); later calls just count."""
        if self._i is None:
            if timestep is None:
                self._i = 0
            else:
                ts = self.timesteps.double().cpu().numpy()
                hit = np.nonzero(np.isclose(ts, float(timestep)))[0]
                if len(hit) == 0:
                    raise ValueError(f"timestep {float(timestep)} is not in the schedule set by set_timesteps")
                self._i = int(hit[1] if len(hit) > 1 else hit[0])
        return self._i

    def set_begin_index(self, begin_index=0):
        self._i = int(begin_index)

    def add_noise_coefficients(self, timestep):
        """(a, b) of add_noise(x0, noise, t) = a x0 + b noise for one timestep (host floats)."""
        t = int(timestep)
        return float(self.ac[t] ** 0.5), float((1 - self.ac[t]) ** 0.5)

    def add_noise(self, original, noise, timesteps):
        t = torch.as_tensor(timesteps).reshape(-1).long().cpu().numpy()
        a = torch.tensor(self.ac[t] ** 0.5, dtype=torch.float32, device=original.device)
        s = torch.tensor((1 - self.ac[t]) ** 0.5, dtype=torch.float32, device=original.device)
        while a.ndim < original.ndim:
            a, s = a.unsqueeze(-1), s.unsqueeze(-1)
        return (a * original.float() + s * noise.float()).to(original.dtype)


class DDIMScheduler(_Base):
    """eta = 0 DDIM."""

    def __init__(self, **kw):
        kw.setdefault("timestep_spacing", "leading")
        super().__init__(**kw)
        self.final_alpha_cumprod = self.ac[0]  # set_alpha_to_one=False

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        self.num_inference_steps = num_inference_steps
        self.timesteps = torch.from_numpy(self._leading(num_inference_steps)).to(device)

    def step_coefficients(self, timestep):
        """x_prev = c_x * x + c_eps * eps (DDIM eta=0 as one affine update)."""
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.ac[t]
        a_prev = self.ac[prev] if prev >= 0 else self.final_alpha_cumprod
        c_x = (a_prev / a_t) ** 0.5
        c_eps = (1 - a_prev) ** 0.5 - (a_prev * (1 - a_t) / a_t) ** 0.5
        return float(c_x), float(c_eps)

    def fused_plan(self, timestep):
        c_x, c_eps = self.step_coefficients(timestep)
        return FusedPlan(in_scale=1.0, c_x=c_x, c_eps=c_eps, c_hist=0.0, h_x=0.0, h_eps=0.0, use_hist=False)

    def fused_commit(self):
        pass

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        c_x, c_eps = self.step_coefficients(timestep)
        prev = (c_x * sample.float() + c_eps * model_output.float()).to(sample.dtype)
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


class DPMSolverMultistepScheduler(_Base):
    """DPM-Solver++(2M), midpoint, lower_order_final, final sigma 0 ("DPM++ 2M" in the registry)."""

    def __init__(self, **kw):
        kw.setdefault("timestep_spacing", "linspace")
        super().__init__(**kw)

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        T = self.config.num_train_timesteps
        if self.config.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        else:
            ts = self._leading(n, extra=1)[:-1]
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts).to(device)
        self._i = None
        self._m_prev = None
        self._fused_hist = False

    @staticmethod
    def _alpha_sigma(s):
        a = 1.0 / (s * s + 1.0) ** 0.5
        return a, s * a

    def fused_plan(self, timestep=None):
        """The arithmetic of `step` below, collected into coefficients of (x, eps, previous x0)."""
        i, n = self._begin(timestep), self.num_inference_steps
        a0, sg0 = self._alpha_sigma(self.sigmas[i])
        a_t, sg_t = self._alpha_sigma(self.sigmas[i + 1])
        lam0 = np.log(a0) - np.log(sg0)
        lam_t = np.log(a_t) - np.log(sg_t) if sg_t > 0 else np.inf
        h = lam_t - lam0
        em1 = float(np.exp(-h) - 1.0)
        h_x, h_eps = 1.0 / a0, -sg0 / a0
        first_order = i == n - 1 or not getattr(self, "_fused_hist", False)
        b, c_hist = -a_t * em1, 0.0
        if not first_order:
            a1, sg1 = self._alpha_sigma(self.sigmas[i - 1])
            r0 = (lam0 - (np.log(a1) - np.log(sg1))) / h
            b = -a_t * em1 * (1.0 + 0.5 / r0)
            c_hist = 0.5 * a_t * em1 / r0
        return FusedPlan(in_scale=1.0, c_x=float(sg_t / sg0 + b * h_x), c_eps=float(b * h_eps), c_hist=float(c_hist),
                         h_x=float(h_x), h_eps=float(h_eps), use_hist=True)

    def fused_commit(self):
        self._fused_hist = True
        self._i += 1

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        i = self._begin(timestep)
        n = self.num_inference_steps
        a0, sg0 = self._alpha_sigma(self.sigmas[i])
        a_t, sg_t = self._alpha_sigma(self.sigmas[i + 1])
        x = sample.float()
        m0 = (x - sg0 * model_output.float()) / a0
        lam0 = np.log(a0) - np.log(sg0)
        lam_t = np.log(a_t) - np.log(sg_t) if sg_t > 0 else np.inf
        h = lam_t - lam0
        em1 = float(np.exp(-h) - 1.0)
        first_order = i == n - 1 or self._m_prev is None
        out = float(sg_t / sg0) * x - float(a_t) * em1 * m0
        if not first_order:
            a1, sg1 = self._alpha_sigma(self.sigmas[i - 1])
            lam1 = np.log(a1) - np.log(sg1)
            r0 = (lam0 - lam1) / h
            d1 = (m0 - self._m_prev) * float(1.0 / r0)
            out = out - 0.5 * float(a_t) * em1 * d1
        self._m_prev = m0
        self._i += 1
        prev = out.to(sample.dtype)
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


class EulerDiscreteScheduler(_Base):
    """The reference's default scheduler (stable_diffusion.py:135-138)."""

    def __init__(self, **kw):
        kw.setdefault("timestep_spacing", "leading")
        super().__init__(**kw)

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        ts = self._leading(n).astype(np.float64)
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig_all)), sig_all)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts.astype(np.float32)).to(device)
        self._i = None

    def scale_model_input(self, sample, timestep=None):
        s = self.sigmas[self._begin(timestep)]
        return (sample.float() / float((s * s + 1) ** 0.5)).to(sample.dtype)

    def add_noise_coefficients(self, timestep):
        ts = self.timesteps.double().cpu().numpy()
        return 1.0, float(self.sigmas[int(np.abs(ts - float(timestep)).argmin())])

    def add_noise(self, original, noise, timesteps):
        """sigma-space noising (diffusers EulerDiscreteScheduler.add_noise): x0 + sigma(t) * noise with
        sigma taken at the schedule position of each timestep (first match, else the nearest)."""
        ts = self.timesteps.double().cpu().numpy()
        want = torch.as_tensor(timesteps).reshape(-1).double().cpu().numpy()
        idx = [int(np.abs(ts - w).argmin()) for w in want]
        s = torch.tensor(self.sigmas[idx], dtype=torch.float32, device=original.device)
        while s.ndim < original.ndim:
            s = s.unsqueeze(-1)
        return (original.float() + s * noise.float()).to(original.dtype)

    def fused_plan(self, timestep=None):
        i = self._begin(timestep)
        s, s_next = self.sigmas[i], self.sigmas[i + 1]
        return FusedPlan(in_scale=float(1.0 / (s * s + 1) ** 0.5), c_x=1.0, c_eps=float(s_next - s), c_hist=0.0,
                         h_x=0.0, h_eps=0.0, use_hist=False)

    def fused_commit(self):
        self._i += 1

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        i = self._begin(timestep)
        s, s_next = self.sigmas[i], self.sigmas[i + 1]
        prev = (sample.float() + model_output.float() * float(s_next - s)).to(sample.dtype)
        self._i += 1
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


class EulerAncestralDiscreteScheduler(EulerDiscreteScheduler):
    """"euler_a" of the registry: Euler step to sigma_down plus fresh noise scaled by sigma_up.  The reference
    calls `step(noise_pred, t, latents)` without a generator (`sd_unified_pipeline.py:489`), so the noise comes
    from torch's global generator on the sample's device; `generator=` / `noise=` are accepted for tests."""

    supports_fused = False                    # stochastic: no affine device step

    def step(self, model_output, timestep, sample, return_dict=False, generator=None, noise=None, **kw):
        i = self._begin(timestep)
        s, s_to = self.sigmas[i], self.sigmas[i + 1]
        s_up = (s_to ** 2 * (s ** 2 - s_to ** 2) / s ** 2) ** 0.5
        s_down = (s_to ** 2 - s_up ** 2) ** 0.5
        x = sample.float()
        prev = x + model_output.float() * float(s_down - s)      # derivative (x - x0) / sigma = eps
        if noise is None:
            noise = torch.randn(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        prev = prev + noise.float() * float(s_up)
        self._i += 1
        prev = prev.to(sample.dtype)
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


def _karras_sigmas(sig_all, n, rho=7.0):
    """diffusers `_convert_to_karras` on the flipped training sigmas: n values from sigma_max down to sigma_min."""
    s_min, s_max = float(sig_all[0]), float(sig_all[-1])
    ramp = np.linspace(0, 1, n)
    lo, hi = s_min ** (1 / rho), s_max ** (1 / rho)
    return (hi + ramp * (lo - hi)) ** rho


def _sigma_to_t(sigma, log_sigmas):
    """diffusers `_sigma_to_t`: fractional training timestep whose log-sigma interpolates to log(sigma)."""
    ls = np.log(np.maximum(sigma, 1e-10))
    dists = ls - log_sigmas[:, None]
    low = np.cumsum(dists >= 0, axis=0).argmax(axis=0).clip(max=len(log_sigmas) - 2)
    high = low + 1
    w = np.clip((log_sigmas[low] - ls) / (log_sigmas[low] - log_sigmas[high]), 0, 1)
    return (1 - w) * low + w * high


class DPMSolverKarrasScheduler(DPMSolverMultistepScheduler):
    """"DPM++ 2M Karras": DPMSolverMultistepScheduler.from_config(config, use_karras_sigmas=True)
    (`models/stable_diffusion.py:213-214`): Karras rho = 7 sigma ladder, timesteps = rounded sigma -> t."""

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        sig = _karras_sigmas(sig_all, n)
        ts = _sigma_to_t(sig, np.log(sig_all)).round().astype(np.int64)
        self.sigmas = np.concatenate([sig, [0.0]])
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts).to(device)
        self._i = None
        self._m_prev = None
        self._fused_hist = False


class DPMSolverSDEScheduler(DPMSolverMultistepScheduler):
    """"DPM++ 2M SDE Karras" of the registry: `from_config(config, se_karras_sigmas=True,
    algorithm_type="sde-dpmsolver++")` (`models/stable_diffusion.py:215-218`).  The keyword is misspelt in the
    reference (`se_karras_sigmas`), so diffusers keeps `use_karras_sigmas=False`: what actually runs is
    sde-dpmsolver++ (2M, midpoint) on the ordinary sigma schedule -- reproduced as it runs, not as it is named."""

    supports_fused = False                    # stochastic

    def step(self, model_output, timestep, sample, return_dict=False, generator=None, noise=None, **kw):
        i = self._begin(timestep)
        n = self.num_inference_steps
        a0, sg0 = self._alpha_sigma(self.sigmas[i])
        a_t, sg_t = self._alpha_sigma(self.sigmas[i + 1])
        x = sample.float()
        m0 = (x - sg0 * model_output.float()) / a0
        lam0 = np.log(a0) - np.log(sg0)
        lam_t = np.log(a_t) - np.log(sg_t) if sg_t > 0 else np.inf
        h = lam_t - lam0
        if noise is None:
            noise = torch.randn(model_output.shape, generator=generator, device=sample.device, dtype=model_output.dtype)
        e1, e2 = float(np.exp(-h)), float(1.0 - np.exp(-2.0 * h))
        out = float(sg_t / sg0 * e1) * x + float(a_t * e2) * m0 + float(sg_t * e2 ** 0.5) * noise.float()
        if not (i == n - 1 or self._m_prev is None):
            a1, sg1 = self._alpha_sigma(self.sigmas[i - 1])
            r0 = (lam0 - (np.log(a1) - np.log(sg1))) / h
            out = out + float(0.5 * a_t * e2 / r0) * (m0 - self._m_prev)
        self._m_prev = m0
        self._i += 1
        prev = out.to(sample.dtype)
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


class PNDMScheduler(_Base):
    """PLMS (PNDM with skip_prk_steps=True, as every SD scheduler config has it): 4th-order linear multistep
    on epsilon; the schedule has num_inference_steps + 1 entries (the second value is visited twice)."""

    def __init__(self, **kw):
        kw.setdefault("timestep_spacing", "leading")
        super().__init__(**kw)
        self.final_alpha_cumprod = self.ac[0]  # set_alpha_to_one=False

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        self.num_inference_steps = n
        base = self._leading(n)[::-1]                                   # ascending
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64)).to(device)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def _prev_sample(self, sample, t, prev_t, eps):
        a_t = self.ac[t]
        a_p = self.ac[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return float(coeff) * sample - float((a_p - a_t) / denom) * eps

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev_t = t - ratio
        eps = model_output.float()
        x = sample.float()
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev_t = t
            t = t + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = x
        elif len(self.ets) == 1 and self.counter == 1:
            eps = (eps + self.ets[-1]) / 2
            x = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            eps = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            eps = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            eps = (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4]) / 24
        prev = self._prev_sample(x, t, prev_t, eps).to(sample.dtype)
        self.counter += 1
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


class UniPCMultistepScheduler(_Base):
    """UniPC (order 2, bh2, predict_x0, lower_order_final): UniP predictor + UniC corrector on the data
    prediction; sigma schedule by linear interpolation, last sigma = the training schedule's smallest."""
    solver_order = 2

    def __init__(self, **kw):
        kw.setdefault("timestep_spacing", "linspace")
        super().__init__(**kw)

    def set_timesteps(self, num_inference_steps, device=None, **kw):
        n = num_inference_steps
        T = self.config.num_train_timesteps
        if self.config.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        else:
            ts = self._leading(n, extra=1)[:-1]
        sig_all = ((1 - self.ac) / self.ac) ** 0.5
        self.sigmas = np.concatenate([np.interp(ts, np.arange(0, len(sig_all)), sig_all), [sig_all[0]]])
        self.num_inference_steps = n
        self.timesteps = torch.from_numpy(ts).to(device)
        self._i = None
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.this_order = 1

    @staticmethod
    def _al(s):
        a = 1.0 / (s * s + 1.0) ** 0.5
        return a, s * a, np.log(a) - np.log(s * a)

    def _bh_terms(self, h, order):
        hh = -h
        h_phi_1 = np.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = np.expm1(hh)
        fact, b = 1, []
        for k in range(1, order + 1):
            b.append(h_phi_k * fact / B_h)
            fact *= k + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return h_phi_1, B_h, np.array(b)

    def _predict(self, sample, order):
        i = self._i
        m0 = self.model_outputs[-1]
        a_t, sg_t, lam_t = self._al(self.sigmas[i + 1])
        a_0, sg_0, lam_0 = self._al(self.sigmas[i])
        h = lam_t - lam_0
        h_phi_1, B_h, _ = self._bh_terms(h, order)
        x_t = float(sg_t / sg_0) * sample - float(a_t * h_phi_1) * m0
        if order == 2:
            _, _, lam_1 = self._al(self.sigmas[i - 1])
            rk = (lam_1 - lam_0) / h
            d1 = (self.model_outputs[-2] - m0) / float(rk)
            x_t = x_t - float(a_t * B_h * 0.5) * d1
        return x_t

    def _correct(self, model_t, last_sample, order):
        i = self._i
        m0 = self.model_outputs[-1]
        a_t, sg_t, lam_t = self._al(self.sigmas[i])
        a_0, sg_0, lam_0 = self._al(self.sigmas[i - 1])
        h = lam_t - lam_0
        h_phi_1, B_h, b = self._bh_terms(h, order)
        x_t = float(sg_t / sg_0) * last_sample - float(a_t * h_phi_1) * m0
        if order == 1:
            rhos = np.array([0.5])
            corr = 0.0
        else:
            _, _, lam_1 = self._al(self.sigmas[i - 2])
            rk = (lam_1 - lam_0) / h
            R = np.array([[1.0, 1.0], [rk, 1.0]])
            rhos = np.linalg.solve(R, b)
            corr = float(rhos[0]) * ((self.model_outputs[-2] - m0) / float(rk))
        return x_t - float(a_t * B_h) * (corr + float(rhos[-1]) * (model_t - m0))

    def step(self, model_output, timestep, sample, return_dict=False, **kw):
        i = self._begin(timestep)
        n = len(self.timesteps)
        x = sample.float()
        a_i, sg_i, _ = self._al(self.sigmas[i])
        x0 = (x - float(sg_i) * model_output.float()) / float(a_i)
        if i > 0 and self.last_sample is not None:
            x = self._correct(x0, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [x0]
        order = min(self.solver_order, n - i)                    # lower_order_final
        self.this_order = min(order, self.lower_order_nums + 1)
        self.last_sample = x
        prev = self._predict(x, self.this_order).to(sample.dtype)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self._i += 1
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev)


REGISTRY = {
    # names of /root/reference/models/stable_diffusion.py:199-227 (all eight)
    "DDIM": lambda cfg: DDIMScheduler.from_config(cfg),
    "euler": lambda cfg: EulerDiscreteScheduler.from_config(cfg),
    "euler_a": lambda cfg: EulerAncestralDiscreteScheduler.from_config(cfg),
    "DPM++ 2M": lambda cfg: DPMSolverMultistepScheduler.from_config(cfg),
    "DPM++ 2M Karras": lambda cfg: DPMSolverKarrasScheduler.from_config(cfg),
    "DPM++ 2M SDE Karras": lambda cfg: DPMSolverSDEScheduler.from_config(cfg),
    "PNDM": lambda cfg: PNDMScheduler.from_config(cfg),
    "uni_pc": lambda cfg: UniPCMultistepScheduler.from_config(cfg),
}
