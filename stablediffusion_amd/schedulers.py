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


REGISTRY = {
    # names of /root/reference/models/stable_diffusion.py:199-227
    "DDIM": lambda cfg: DDIMScheduler.from_config(cfg),
    "euler": lambda cfg: EulerDiscreteScheduler.from_config(cfg),
    "DPM++ 2M": lambda cfg: DPMSolverMultistepScheduler.from_config(cfg),
}
