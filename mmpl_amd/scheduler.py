"""Flow-matching samplers of the denoise loop, MI355X host side.

``FlowUniPCMultistepScheduler`` keeps the reference's name and call shape
(MMPL_t2v/wan/utils/fm_solvers_unipc.py:20-800 as configured by
pipeline/casual_fps_inference.py:503-511: order 2, bh2, predict_x0, flow_prediction, lower_order_final,
final sigma 0) but splits the work the MI355X way: the per-step *scalars* (sigma ratios, h, expm1 terms,
rho coefficients -- a handful of fp32 values) are computed on the host exactly as the reference computes
them, and the whole tensor update (CFG combine, x0 conversion, UniC corrector, UniP predictor, solver-state
rotation) is ONE fused HIP kernel (``mmpl_cfg_unipc_step``) instead of ~25 tiny PyTorch launches.

``FlowMatchScheduler`` is the train-time schedule the pipeline uses for ``add_noise``
(MMPL_t2v/utils/scheduler.py:103-176).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from . import _lib


class FlowUniPCMultistepScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, solver_order: int = 2, shift: float = 1.0,
                 use_dynamic_shifting: bool = False):
        assert solver_order == 2 and not use_dynamic_shifting
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = torch.from_numpy(1.0 - alphas).to(dtype=torch.float32)
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigmas = sigmas
        self.sigma_min = self.sigmas[-1].item()
        self.sigma_max = self.sigmas[0].item()
        self.shift = shift
        self.timesteps = sigmas * num_train_timesteps
        self.num_inference_steps = None
        self._state = None

    def set_timesteps(self, num_inference_steps: int, device=None, shift: Optional[float] = None):
        sigmas = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1).copy()[:-1]
        if shift is None:
            shift = self.shift
        sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
        timesteps = sigmas * self.num_train_timesteps
        sigmas = np.concatenate([sigmas, [0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)                       # stays on the host, like the reference
        self.timesteps = torch.from_numpy(timesteps).to(dtype=torch.int64)
        self.num_inference_steps = len(timesteps)
        self.lower_order_nums = 0
        self.step_index = 0
        self.this_order = 1
        self._have_last = False
        self._state = None

    # -- host scalars ------------------------------------------------------------------------------
    @staticmethod
    def _lam(sigma):
        return torch.log(1 - sigma) - torch.log(sigma)

    def _corrector_scalars(self, order: int):
        """fm_solvers_unipc.py:549-618."""
        si = self.step_index
        sigma_t, sigma_s0 = self.sigmas[si], self.sigmas[si - 1]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        rks = []
        inv_rk = 0.0
        if order == 2:
            rk = (self._lam(self.sigmas[si - 2]) - self._lam(sigma_s0)) / h
            rks.append(rk)
            inv_rk = (1.0 / rk).item()
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
        if order == 1:
            rhos = torch.tensor([0.5], dtype=torch.bfloat16)
        else:
            rhos = torch.linalg.solve(torch.stack(R), torch.tensor(b)).to(torch.bfloat16)
        return dict(c1=(sigma_t / sigma_s0).item(), c2=(alpha_t * h_phi_1).item(), c3=(alpha_t * B_h).item(), inv_rk=inv_rk,
                    rho0=rhos[0].float().item() if order == 2 else 0.0, rho_last=rhos[-1].float().item())

    def _predictor_scalars(self, order: int):
        """fm_solvers_unipc.py:402-476."""
        si = self.step_index
        sigma_t, sigma_s0 = self.sigmas[si + 1], self.sigmas[si]
        alpha_t = 1 - sigma_t
        h = self._lam(sigma_t) - self._lam(sigma_s0)
        inv_rk = 0.0
        if order == 2:
            rk = (self._lam(self.sigmas[si - 1]) - self._lam(sigma_s0)) / h
            inv_rk = (1.0 / rk).item()
        hh = -h
        h_phi_1 = torch.expm1(hh)
        B_h = torch.expm1(hh)
        return dict(c1=(sigma_t / sigma_s0).item(), c2=(alpha_t * h_phi_1).item(), c3=(alpha_t * B_h).item(), inv_rk=inv_rk)

    def step_scalars(self, guidance: float) -> _lib.MmplUniPCStep:
        """Scalars of the step at the current step_index; advances the host-side order bookkeeping (:686-730)."""
        si = self.step_index
        use_corr = si > 0 and self._have_last
        st = _lib.MmplUniPCStep()
        st.guidance = float(guidance)
        st.sigma_cur = self.sigmas[si].item()
        st.use_corrector = int(use_corr)
        st.corr_order = self.this_order
        if use_corr:
            c = self._corrector_scalars(self.this_order)
            st.c_c1, st.c_c2, st.c_c3, st.c_inv_rk, st.c_rho0, st.c_rho_last = c["c1"], c["c2"], c["c3"], c["inv_rk"], c["rho0"], c["rho_last"]
        this_order = min(self.solver_order, len(self.timesteps) - si)
        self.this_order = min(this_order, self.lower_order_nums + 1)
        p = self._predictor_scalars(self.this_order)
        st.pred_order = self.this_order
        st.p_c1, st.p_c2, st.p_c3, st.p_inv_rk = p["c1"], p["c2"], p["c3"], p["inv_rk"]
        self._have_last = True
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return st

    # -- device update -----------------------------------------------------------------------------
    def _ensure_state(self, sample: torch.Tensor):
        if self._state is None or self._state[0].shape != sample.shape or self._state[0].device != sample.device:
            self._state = [torch.zeros_like(sample) for _ in range(3)]      # m0, m1, last_sample

    def step_cfg(self, flow_cond: torch.Tensor, flow_uncond: Optional[torch.Tensor], guidance: float,
                 sample: torch.Tensor) -> torch.Tensor:
        """CFG combine + scheduler step, fused; `sample` is updated in place and returned."""
        assert sample.is_contiguous() and sample.dtype == torch.bfloat16 and flow_cond.is_contiguous()
        self._ensure_state(sample)
        st = self.step_scalars(guidance)
        lib = _lib.load()
        m0, m1, last = self._state
        _lib.check(lib.mmpl_cfg_unipc_step(_lib.ptr(flow_cond), _lib.ptr(flow_uncond), _lib.ptr(sample), _lib.ptr(m0),
                                           _lib.ptr(m1), _lib.ptr(last), sample.numel(), C.byref(st), _lib.stream_ptr()),
                   "mmpl_cfg_unipc_step")
        return sample

    # -- device-resident step table: one hipGraph per denoise step, replayed with no host work in between --------------
    def build_step_table(self, guidance: float, device) -> None:
        """Upload the scalars of ALL remaining steps (they depend only on the step index, never on data) plus the timestep
        of every step; `step_cfg_table` then reads entry *counter on the device.  The host-side step bookkeeping of THIS
        object is not advanced (the scalars come from a copy): after the table's last step the device kernel does nothing
        (the counter is clamped), `reset_step_table` rewinds it for another pass over the same schedule."""
        import copy
        probe = copy.copy(self)
        probe._state = None
        n = len(self.timesteps) - self.step_index
        rows = (_lib.MmplUniPCStep * n)()
        for i in range(n):
            rows[i] = probe.step_scalars(guidance)
        raw = torch.frombuffer(bytearray(bytes(rows)), dtype=torch.uint8).clone()
        self._table = raw.to(device)
        t_host = [float(t) for t in self.timesteps[self.step_index:]]
        self._t_table = torch.tensor(t_host, dtype=torch.float32, device=device)
        self._t_first = t_host[0]          # host copy: reset_step_table must not read the device table (a sync inside timed loops)
        self._counter = torch.zeros(1, dtype=torch.int32, device=device)
        self._table_n = n

    def reset_step_table(self, timestep: torch.Tensor) -> None:
        """Rewind the device step counter, the solver history and `timestep` to the first entry of the uploaded table."""
        self._counter.zero_()
        if self._state is not None:
            for t in self._state:
                t.zero_()
        timestep.fill_(self._t_first)

    def step_cfg_table(self, flow_cond: torch.Tensor, flow_uncond: torch.Tensor, sample: torch.Tensor, timestep: torch.Tensor) -> None:
        """CFG combine + scheduler step with device-resident scalars (capturable: no host value enters the launch);
        advances the device step counter and writes the next step's timestep into `timestep` (float32, contiguous)."""
        assert sample.is_contiguous() and sample.dtype == torch.bfloat16 and timestep.dtype == torch.float32 and timestep.is_contiguous()
        self._ensure_state(sample)
        m0, m1, last = self._state
        _lib.check(_lib.load().mmpl_cfg_unipc_step_table(
            _lib.ptr(flow_cond), _lib.ptr(flow_uncond), _lib.ptr(sample), _lib.ptr(m0), _lib.ptr(m1), _lib.ptr(last), sample.numel(),
            _lib.ptr(self._table), _lib.ptr(self._counter), _lib.ptr(timestep), _lib.ptr(self._t_table), timestep.numel(), self._table_n,
            _lib.stream_ptr()), "mmpl_cfg_unipc_step_table")

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        """Reference call shape (fm_solvers_unipc.py:655).  `timestep` is accepted and ignored like the reference's
        internal step counter does after the first call."""
        out = self.step_cfg(model_output.contiguous(), None, 0.0, sample.contiguous().clone())
        return (out,) if not return_dict else {"prev_sample": out}


class FlowMatchScheduler:
    """utils/scheduler.py:103-176 (shift, sigma_min, extra_one_step) -- host-side, tiny."""

    def __init__(self, num_inference_steps=100, num_train_timesteps=1000, shift=3.0, sigma_max=1.0, sigma_min=0.003 / 1.002,
                 extra_one_step=False):
        self.num_train_timesteps, self.shift, self.sigma_max, self.sigma_min = num_train_timesteps, shift, sigma_max, sigma_min
        self.extra_one_step = extra_one_step
        self.set_timesteps(num_inference_steps)

    def set_timesteps(self, num_inference_steps=100, denoising_strength=1.0, training=False):
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            self.sigmas = torch.linspace(sigma_start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            self.sigmas = torch.linspace(sigma_start, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * self.sigmas / (1 + (self.shift - 1) * self.sigmas)
        self.timesteps = self.sigmas * self.num_train_timesteps

    def add_noise(self, original_samples, noise, timestep):
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        sig = self.sigmas.to(noise.device)
        ts = self.timesteps.to(noise.device)
        tid = torch.argmin((ts.unsqueeze(0) - timestep.to(noise.device).unsqueeze(1)).abs(), dim=1)
        sigma = sig[tid].reshape(-1, 1, 1, 1)
        return ((1 - sigma) * original_samples + sigma * noise).type_as(noise)
