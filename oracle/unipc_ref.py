"""Oracle: FlowUniPCMultistepScheduler (order 2, bh2, predict_x0, lower_order_final) and
FlowMatchScheduler.add_noise, restated.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  * MMPL_t2v/wan/utils/fm_solvers_unipc.py:76-132 (ctor), :160-229 (set_timesteps),
    :280-331 (convert_model_output), :350-484 (UniP), :486-626 (UniC), :655-739 (step)
  * MMPL_t2v/utils/scheduler.py:103-140 (FlowMatchScheduler.set_timesteps), :159-176 (add_noise)
as used by MMPL_t2v/pipeline/casual_fps_inference.py:503-511 (num_train_timesteps=1000, shift=1 in the
ctor, set_timesteps(50, shift=5.0)).  Tensors keep their dtype (bf16 on the real path); the scalar
coefficients are 0-dim fp32 CPU tensors exactly as in the reference, so every tensor op rounds where
the reference rounds.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


class FlowUniPCRef:
    """`gpu_scalar_semantics`: PyTorch's *CPU* kernels round a 0-dim fp32 tensor to the tensor dtype (bf16) when it
    is the FIRST operand of a multiply (`sigma_t * x`), while its GPU kernels keep the scalar in fp32 on either
    side (opmath_symmetric_gpu_kernel_with_scalars).  The reference's native platform is the GPU, so the HIP step
    follows the GPU semantics; with this flag the oracle writes the same products tensor-first (`x * sigma_t`),
    which reproduces the GPU semantics on the CPU.  Flag off (default) = the literal reference expression order,
    which is what the golden fixture (reference run on CPU) pins bit-exactly."""

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, shift: float = 1.0,
                 gpu_scalar_semantics: bool = False):
        self.gpu_scalar_semantics = gpu_scalar_semantics
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                  # :112-116
        self.sigmas = sigmas
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.shift = shift
        self.timesteps = sigmas * num_train_timesteps

    def set_timesteps(self, num_inference_steps: int, shift: Optional[float] = None):
        sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]   # :182-185
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)                                         # :190-194
        timesteps = sigmas * self.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)                                   # final_sigmas_type "zero"
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * self.solver_order
        self.lower_order_nums = 0
        self.last_sample = None
        self.step_index = 0
        self.this_order = 1

    def _sm(self, scalar, tensor):
        """scalar-tensor product in the reference's operand order, or tensor-first for GPU scalar semantics."""
        if isinstance(tensor, (int, float)):
            return scalar * tensor
        return tensor * scalar if self.gpu_scalar_semantics else scalar * tensor

    # :315-331
    def _convert(self, model_output, sample):
        sigma_t = self.sigmas[self.step_index]
        return sample - self._sm(sigma_t, model_output)

    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    # :350-484
    def _uni_p(self, sample, order):
        m0 = self.model_outputs[-1]
        x = sample
        sigma_t, sigma_s0 = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        D1s = []
        for i in range(1, order):
            si = self.step_index - i
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lam(self.sigmas[si]) - self._lam(sigma_s0)) / h
            D1s.append((mi - m0) / rk)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        x_t_ = self._sm(sigma_t / sigma_s0, x) - self._sm(alpha_t * h_phi_1, m0)
        if D1s:
            D1s = torch.stack(D1s, dim=1)
            rhos_p = torch.tensor([0.5], dtype=x.dtype)          # order == 2 simplified branch
            pred_res = torch.einsum("k,bkc...->bc...", rhos_p, D1s)
        else:
            pred_res = 0
        x_t = x_t_ - self._sm(alpha_t * B_h, pred_res)
        return x_t.to(x.dtype)

    # :486-626
    def _uni_c(self, this_model_output, last_sample, this_sample, order):
        m0 = self.model_outputs[-1]
        x = last_sample
        model_t = this_model_output
        sigma_t, sigma_s0 = self.sigmas[self.step_index], self.sigmas[self.step_index - 1]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        rks, D1s = [], []
        for i in range(1, order):
            si = self.step_index - (i + 1)
            mi = self.model_outputs[-(i + 1)]
            rk = (self._lam(self.sigmas[si]) - self._lam(sigma_s0)) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(1.0)
        rks = torch.tensor(rks)
        hh = -h
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)
        factorial_i = 1
        R, b = [], []
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        R = torch.stack(R)
        b = torch.tensor(b)
        if order == 1:
            rhos_c = torch.tensor([0.5], dtype=x.dtype)
        else:
            rhos_c = torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = self._sm(sigma_t / sigma_s0, x) - self._sm(alpha_t * h_phi_1, m0)
        if D1s:
            corr_res = torch.einsum("k,bkc...->bc...", rhos_c[:-1], torch.stack(D1s, dim=1))
        else:
            corr_res = 0
        D1_t = model_t - m0
        x_t = x_t_ - self._sm(alpha_t * B_h, corr_res + rhos_c[-1] * D1_t)
        return x_t.to(x.dtype)

    # :655-739
    def step(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        use_corrector = self.step_index > 0 and self.last_sample is not None
        m_conv = self._convert(model_output, sample)
        if use_corrector:
            sample = self._uni_c(m_conv, self.last_sample, sample, self.this_order)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = m_conv
        this_order = min(self.solver_order, len(self.timesteps) - self.step_index)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._uni_p(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev


class FlowMatchRef:
    """utils/scheduler.py:103-140,159-176 with (shift, sigma_min=0, extra_one_step=True), 1000 training steps."""

    def __init__(self, shift: float = 5.0, num_train_timesteps: int = 1000):
        sig = torch.linspace(1.0, 0.0, num_train_timesteps + 1)[:-1]
        self.sigmas = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = self.sigmas * num_train_timesteps

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        tid = torch.argmin((self.timesteps.unsqueeze(0) - timestep.unsqueeze(1)).abs(), dim=1)
        sigma = self.sigmas[tid].reshape(-1, 1, 1, 1)
        return ((1 - sigma) * original + sigma * noise).type_as(noise)
