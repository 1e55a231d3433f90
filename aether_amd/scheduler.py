"""Host-side stand-in for diffusers' `CogVideoXDPMScheduler` as the reference drives it
(/root/reference/scripts/demo.py:220-222 builds it; /root/reference/aether/pipelines/aetherv1_pipeline_cogvideox.py
uses `set_timesteps` via P:227, `scale_model_input` P:835, `step` P:907-915, `init_noise_sigma` P:686, `order` P:821).

The schedule is a few hundred float64 scalars and the update is element-wise over 3.3 M latent values per step, so
it stays host-driven PyTorch — what matters for parity is that the random draws (count, order,
shape, dtype, device of every `randn`) and the dtype of every intermediate are those of the reference: scalars are
0-dim float64 CPU tensors exactly like diffusers' `alphas_cumprod[t]`, so PyTorch's type promotion rounds the same
products to bf16 / fp32 as it does there.  Algorithm restated from diffusers 0.32 schedulers/scheduling_dpm_cogvideox.py
(SDE DPM-Solver++(2M), v-prediction, zero-terminal-SNR + SNR-shifted scaled-linear betas) — UPSTREAM-UNVERIFIED.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.torch_utils.randn_tensor: draw on the generator's device, then move (P:683)."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    if isinstance(generator, list):
        shape = (1,) + tuple(shape[1:])
        return torch.cat([randn_tensor(shape, g, device, dtype) for g in generator], dim=0)
    rand_device = device
    if generator is not None:
        gen_type = generator.device.type
        if gen_type != device.type and gen_type == "cpu":
            rand_device = torch.device("cpu")
        elif gen_type != device.type and gen_type == "cuda":
            raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gen_type}.")
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype).to(device)


def _rescale_zero_terminal_snr(alphas_cumprod: torch.Tensor) -> torch.Tensor:
    a = alphas_cumprod.sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = (a - aT) * (a0 / (a0 - aT))
    return a ** 2


class CogVideoXDPMScheduler:
    """NOTE on defaults: a bare `CogVideoXDPMScheduler()` takes the values of THUDM/CogVideoX-5b's `scheduler/scheduler_config.json`
    (what `from_pretrained` reads at /root/reference/scripts/demo.py:220-222: `snr_shift_scale` 1.0, `timestep_spacing`
    "trailing", `prediction_type` "v_prediction", `rescale_betas_zero_snr` True), NOT diffusers' class-level defaults
    (3.0, "leading", "epsilon", False).  Pass them explicitly to get the latter.  `trained_betas` is not supported (raises)."""
    order = 1
    _defaults = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                     clip_sample=False, set_alpha_to_one=True, steps_offset=0, prediction_type="v_prediction",
                     timestep_spacing="trailing", rescale_betas_zero_snr=True, snr_shift_scale=1.0)

    def __init__(self, **config):
        if config.get("trained_betas") is not None:
            raise NotImplementedError("aether_amd: CogVideoXDPMScheduler(trained_betas=...) is not implemented")
        cfg = dict(self._defaults)
        cfg.update({k: v for k, v in config.items() if k in cfg})
        self.config = SimpleNamespace(**cfg)
        c = self.config
        if c.beta_schedule == "scaled_linear":
            betas = torch.linspace(c.beta_start ** 0.5, c.beta_end ** 0.5, c.num_train_timesteps, dtype=torch.float64) ** 2
        elif c.beta_schedule == "linear":
            betas = torch.linspace(c.beta_start, c.beta_end, c.num_train_timesteps, dtype=torch.float32)
        else:
            raise NotImplementedError(f"{c.beta_schedule} is not implemented for {self.__class__}")
        self.betas = betas
        self.alphas = 1.0 - betas
        ac = torch.cumprod(self.alphas, dim=0)
        ac = ac / (c.snr_shift_scale + (1 - c.snr_shift_scale) * ac)
        if c.rescale_betas_zero_snr:
            ac = _rescale_zero_terminal_snr(ac)
        self.alphas_cumprod = ac
        self.final_alpha_cumprod = torch.tensor(1.0) if c.set_alpha_to_one else ac[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, c.num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = "scheduler", **_):
        root = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(root, "scheduler_config.json")) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    @classmethod
    def from_config(cls, config, **kw):
        return cls(**(dict(config) if isinstance(config, dict) else vars(config)), **kw)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        c = self.config
        if num_inference_steps > c.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                             f" {c.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        if c.timestep_spacing == "linspace":
            ts = np.linspace(0, c.num_train_timesteps - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif c.timestep_spacing == "leading":
            ratio = c.num_train_timesteps // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + c.steps_offset
        elif c.timestep_spacing == "trailing":
            ratio = c.num_train_timesteps / num_inference_steps
            ts = np.round(np.arange(c.num_train_timesteps, 0, -ratio)).astype(np.int64) - 1
        else:
            raise ValueError(f"{c.timestep_spacing} is not supported.")
        self.timesteps = torch.from_numpy(ts).to(device)

    @staticmethod
    def _lambda(a):
        return ((a / (1 - a)) ** 0.5).log()

    def _coefficients(self, timestep, timestep_back, second_order_possible: bool):
        """The float64 host scalars of one update (diffusers' `step` expressions, in its order) — used by `step` and `step_fused` alike."""
        c = self.config
        t = int(timestep)
        prev_t = t - c.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        lamb, lamb_next = self._lambda(a_t), self._lambda(a_prev)
        h = lamb_next - lamb
        out = dict(a_sqrt=a_t ** 0.5, b_sqrt=(1 - a_t) ** 0.5, m1=((1 - a_prev) / (1 - a_t)) ** 0.5 * (-h).exp(),
                   m2=(-2 * h).expm1() * a_prev ** 0.5, m_noise=(1 - a_prev) ** 0.5 * (1 - (-2 * h).exp()) ** 0.5,
                   second=bool(second_order_possible and prev_t >= 0))
        if out["second"]:
            r = (lamb - self._lambda(self.alphas_cumprod[int(timestep_back)])) / h
            out["m3"], out["m4"] = 1 + 1 / (2 * r), 1 / (2 * r)
        return out

    def step_fused(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, guidance_scale=None, generator=None):
        """The element-wise tail of one denoise step as ONE HIP kernel (`aether_dpm_step`, csrc/sampler.hip): what the reference's
        loop does between the transformer call and the next iteration (P:876-916) — `noise_pred.float()`, the classifier-free-guidance
        combine when `model_output` holds (unconditional, conditional), `step(...)` and the cast back to the latents' dtype.
        model_output: the transformer's bf16 output [1 or 2, ...]; sample: bf16 latents [1, ...].  Returns (latents bf16, x0 fp32),
        BIT-IDENTICAL to the PyTorch sequence (tests/test_kernels_gpu.py::test_dpm_step_fused_is_bit_identical); the random draws are
        made here with torch, in `step`'s order (one draw, or two when the second-order form applies — the first one is then unused,
        exactly as in diffusers).  v-prediction on a CUDA device only; callers fall back to `step` otherwise."""
        from . import _lib
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.config.prediction_type != "v_prediction" or not sample.is_cuda or sample.dtype != torch.bfloat16 or model_output.dtype != torch.bfloat16:
            raise NotImplementedError("step_fused: v-prediction, bf16 CUDA tensors only")
        nb = model_output.shape[0]
        if sample.shape[0] != 1 or nb not in (1, 2) or (nb == 2) != (guidance_scale is not None):
            raise ValueError("step_fused: sample must be [1, ...], model_output [1, ...] or [2, ...] with a guidance scale")
        k = self._coefficients(timestep, timestep_back, old_pred_original_sample is not None)
        noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        if k["second"]:
            noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        mo, smp = model_output.contiguous(), sample.contiguous()
        old = old_pred_original_sample.contiguous() if k["second"] else None
        x0 = torch.empty(sample.shape, dtype=torch.float32, device=sample.device)
        prev = torch.empty_like(smp)
        f = lambda v: float(torch.as_tensor(v, dtype=torch.float64).to(torch.float32))  # noqa: E731  (the rounding PyTorch applies to a scalar operand)
        _lib.check(_lib.load().aether_dpm_step(mo.data_ptr(), nb, f(guidance_scale if nb == 2 else 0.0), smp.data_ptr(), _lib.ptr(old), noise.data_ptr(),
                                               f(k["a_sqrt"]), f(k["b_sqrt"]), f(k["m1"]), f(k["m2"]), f(k["m_noise"]), f(k.get("m3", 0.0)),
                                               f(k.get("m4", 0.0)), x0.data_ptr(), None, prev.data_ptr(), smp.numel(), _lib.current_stream()),
                   "aether_dpm_step")
        return prev, x0

    def step(self, model_output, old_pred_original_sample, timestep, timestep_back, sample, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = False):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        c = self.config
        k = self._coefficients(timestep, timestep_back, old_pred_original_sample is not None)     # the ONE place the schedule scalars come from
        if c.prediction_type == "epsilon":
            x0 = (sample - k["b_sqrt"] * model_output) / k["a_sqrt"]
        elif c.prediction_type == "sample":
            x0 = model_output
        elif c.prediction_type == "v_prediction":
            x0 = k["a_sqrt"] * sample - k["b_sqrt"] * model_output
        else:
            raise ValueError(f"prediction_type given as {c.prediction_type} must be one of `epsilon`, `sample`, or `v_prediction`")
        noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        prev_sample = k["m1"] * sample - k["m2"] * x0 + k["m_noise"] * noise
        if not k["second"]:
            return (prev_sample, x0)
        d = k["m3"] * x0 - k["m4"] * old_pred_original_sample
        noise = randn_tensor(sample.shape, generator=generator, device=sample.device, dtype=sample.dtype)
        prev_sample = k["m1"] * sample - k["m2"] * d + k["m_noise"] * noise
        return (prev_sample, x0)
