"""Host-side flow-matching scheduler (scalar math on CPU fp32 tensors, like the reference's).

API mirror of DiffSynth-Studio/diffsynth/schedulers/flow_match.py (FlowMatchScheduler:5-125): same
constructor keywords, `set_timesteps`, `step`, `add_noise`, `return_to_timestep`, `sigmas`,
`timesteps`.  The sampler's tensor work (CFG combine + Euler update) is done on the GPU by
`pe_cfg_euler_step`; `step()` here is kept for API parity and for CPU tensors.  The reference's
training-only members (per-timestep loss weights, `training_target`) belong to its trainer, which is
outside this path: they are not mirrored, and `set_timesteps(training=True)` says so.
"""
from __future__ import annotations

import math
from typing import Optional

import torch


class FlowMatchScheduler:
    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0,
                 sigma_min=0.003 / 1.002, inverse_timesteps=False, extra_one_step=False, reverse_sigmas=False,
                 exponential_shift=False, exponential_shift_mu=None, shift_terminal=None):
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.sigma_max = sigma_max
        self.sigma_min = sigma_min
        self.inverse_timesteps = inverse_timesteps
        self.extra_one_step = extra_one_step
        self.reverse_sigmas = reverse_sigmas
        self.exponential_shift = exponential_shift
        self.exponential_shift_mu = exponential_shift_mu
        self.shift_terminal = shift_terminal
        self.set_timesteps(num_inference_steps)

    # -- table construction (flow_match.py:34-69) ------------------------------------------------
    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False, shift=None,
                      dynamic_shift_len=None, exponential_shift_mu=None):
        if training:
            raise ValueError("FlowMatchScheduler.set_timesteps(training=True): the trainer's timestep weights are not part of the "
                             "inference path this package replaces")
        if shift is not None:
            self.shift = shift
        start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        n = num_inference_steps + 1 if self.extra_one_step else num_inference_steps
        s = torch.linspace(start, self.sigma_min, n)
        if self.extra_one_step:
            s = s[:-1]
        if self.inverse_timesteps:
            s = torch.flip(s, dims=[0])
        if self.exponential_shift:
            if exponential_shift_mu is not None:
                mu = exponential_shift_mu
            elif dynamic_shift_len is not None:
                mu = self.calculate_shift(dynamic_shift_len)
            else:
                mu = self.exponential_shift_mu
            e = math.exp(mu)
            s = e / (e + (1 / s - 1))
        else:
            s = self.shift * s / (1 + (self.shift - 1) * s)
        if self.shift_terminal is not None:
            rem = 1 - s
            s = 1 - (rem / (rem[-1] / (1 - self.shift_terminal)))
        if self.reverse_sigmas:
            s = 1 - s
        self.sigmas = s
        self.timesteps = s * self.num_train_timesteps

    # -- lookups -----------------------------------------------------------------------------------
    def _index_of(self, timestep) -> int:
        if isinstance(timestep, torch.Tensor):
            timestep = timestep.cpu()
        return int(torch.argmin((self.timesteps - timestep).abs()))

    def sigma_pair(self, index: int, to_final: bool = False):
        """(sigma_i, sigma_next) as 0-dim fp32 tensors / python ints, as `step` uses them."""
        sigma = self.sigmas[index]
        if to_final or index + 1 >= len(self.timesteps):
            nxt = 1 if (self.inverse_timesteps or self.reverse_sigmas) else 0
        else:
            nxt = self.sigmas[index + 1]
        return sigma, nxt

    def dsigma(self, index: int) -> float:
        """fp32 value of (sigma_next - sigma_i): the scalar the Euler update multiplies by."""
        sigma, nxt = self.sigma_pair(index)
        return float((nxt - sigma).item())

    # -- tensor ops (flow_match.py:72-106) ---------------------------------------------------------
    def step(self, model_output, timestep, sample, to_final=False, **kwargs):
        sigma, nxt = self.sigma_pair(self._index_of(timestep), to_final)
        return sample + model_output * (nxt - sigma)

    def return_to_timestep(self, timestep, sample, sample_stablized):
        return (sample - sample_stablized) / self.sigmas[self._index_of(timestep)]

    def add_noise(self, original_samples, noise, timestep):
        sigma = self.sigmas[self._index_of(timestep)]
        return (1 - sigma) * original_samples + sigma * noise

    @staticmethod
    def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 8192, base_shift: float = 0.5,
                        max_shift: float = 0.9):
        m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
        return image_seq_len * m + (base_shift - m * base_seq_len)


def qwen_image_scheduler() -> FlowMatchScheduler:
    """The instance QwenImagePhysicPipeline.__init__ builds (pipelines/qwen_image_physical.py:192)."""
    return FlowMatchScheduler(sigma_min=0, sigma_max=1, extra_one_step=True, exponential_shift=True,
                              exponential_shift_mu=0.8, shift_terminal=0.02)


# ---------------------------------------------------------------------------------------------------
# per-step host scalars the kernels consume
# ---------------------------------------------------------------------------------------------------
def timestep_sinusoid(t_scaled: torch.Tensor) -> torch.Tensor:
    """TemporalTimesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000,
    align_dtype_to_timestep=True) -- models/utils.py:189-216.  `t_scaled` = timestep/1000 in the
    pipeline dtype, shape [n]; returns fp32 [n,256] ([cos | sin])."""
    half = 128
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000) / half))
    freqs = freqs.to(t_scaled.dtype)                       # frequency table rounded to the timestep dtype
    ang = 1000 * (t_scaled[:, None].float() * freqs[None, :])
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def adapter_alpha(t: torch.Tensor, t_min: float, t_max: float, pred_dtype=torch.bfloat16):
    """VisualThinkingDualAdapter._get_alpha + `.type_as(pred)` (pipelines/helpers.py:142-150,158):
    returns (alpha, 1-alpha) as python floats holding the `pred_dtype`-rounded values the mix uses.
    `t` is the [1] timestep tensor in the pipeline dtype (bf16), so every op here rounds like the
    reference's."""
    a = ((t - t_min) / (t_max - t_min + 1e-6)).clamp(0.0, 1.0).view(-1, 1, 1).to(pred_dtype)
    return float(a.float().item()), float((1 - a).float().item())
