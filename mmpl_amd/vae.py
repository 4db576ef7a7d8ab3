"""Host-side handle of the HIP Wan 3D-VAE (libmmpl_hip.so: mmpl_vae_*), the engine behind ``WanVAEWrapper``.

Takes the reference's ``Wan2.1_VAE.pth`` state_dict as is (MMPL_t2v/wan/modules/vae.py:612-636), repacks the conv
weights once into the implicit-GEMM operand layout ([Cout, taps * Cin], Cin contiguous) and keeps them on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib


class VaeEngine:
    def __init__(self, lat_h: int, lat_w: int, device="cuda:0"):
        self.lat_h, self.lat_w = lat_h, lat_w
        self.device = torch.device(device)
        self._lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mmpl_vae_create(lat_h, lat_w, C.byref(h)), "mmpl_vae_create")
        self._h = h
        self._weights: List[torch.Tensor] = []
        self._ws: Dict[int, torch.Tensor] = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mmpl_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def clear_cache(self):
        """API parity with WanVAE_.clear_cache (vae.py:602-609): every decode/encode call starts from a cleared cache."""

    @staticmethod
    def _repack(name: str, t: torch.Tensor) -> torch.Tensor:
        t = t.to(torch.bfloat16)
        if name.endswith(".gamma"):
            return t.reshape(-1)
        if name.endswith(".bias"):
            if name == "decoder.head.2.bias":
                t = torch.cat([t, t.new_zeros(1)])
            return t.reshape(-1)
        if name in ("conv1.weight", "conv2.weight"):
            return t.reshape(t.shape[0], t.shape[1])
        if t.dim() == 5:
            t = t.permute(0, 2, 3, 4, 1)
        elif t.dim() == 4:
            t = t.permute(0, 2, 3, 1)
        cout, cin = t.shape[0], t.shape[-1]
        t = t.reshape(cout, -1, cin)
        if cin % 32:
            pad = 32 - cin % 32
            t = torch.cat([t, t.new_zeros(cout, t.shape[1], pad)], dim=2)
        t = t.reshape(cout, -1)
        if name == "decoder.head.2.weight":
            t = torch.cat([t, t.new_zeros(1, t.shape[1])])
        return t

    @staticmethod
    def _frag_pack(w2d: torch.Tensor, cin_pad: int) -> torch.Tensor:
        """[N, ntaps * Cin] (tap-major, Cin a multiple of 32) -> the fragment-major packing conv_halo_kernel loads: for every
        (32-channel chunk, tap, 16-row group) the 64 lanes' 16 bytes back to back, lane = 16 * (8-channel k chunk) + row."""
        n, k = w2d.shape
        ntaps, nchunk = k // cin_pad, cin_pad // 32
        npad = (n + 15) // 16 * 16
        if npad != n:
            w2d = torch.cat([w2d, w2d.new_zeros(npad - n, k)])
        return w2d.view(npad // 16, 16, ntaps, nchunk, 4, 8).permute(3, 2, 0, 4, 1, 5).contiguous().view(-1)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        n = self._lib.mmpl_vae_num_weights()
        ws = []
        for i in range(n):
            name = self._lib.mmpl_vae_weight_name(i).decode()
            if name.endswith(".weight.frag"):
                base = name[:-len(".frag")]
                w2d = self._repack(base, sd[base])
                cin = sd[base].shape[1]
                ws.append(self._frag_pack(w2d, (cin + 31) // 32 * 32).to(self.device))
                continue
            ws.append(self._repack(name, sd[name]).contiguous().to(self.device))
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in ws])
        _lib.check(self._lib.mmpl_vae_bind_weights(self._h, arr, n), "mmpl_vae_bind_weights")
        self._weights = ws

    def _workspace(self, mode: int) -> torch.Tensor:
        if mode not in self._ws:
            nbytes = self._lib.mmpl_vae_workspace_bytes(self._h, mode)
            other = self._ws.get(1 - mode)
            if other is not None and other.numel() >= nbytes:
                self._ws[mode] = other
            else:
                self._ws[mode] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws[mode]

    @staticmethod
    def _scales(mean: Sequence[float], std: Sequence[float]):
        m = torch.tensor(mean, dtype=torch.float32).to(torch.bfloat16)
        inv = 1.0 / torch.tensor(std, dtype=torch.float32).to(torch.bfloat16)          # bf16 arithmetic like the wrapper
        return (C.c_float * 16)(*m.float().tolist()), (C.c_float * 16)(*inv.float().tolist())

    def decode(self, latent: torch.Tensor, mean, std) -> torch.Tensor:
        """latent [F, 16, h, w] -> float32 [1 + 4(F-1), 3, 8h, 8w] in [-1, 1]."""
        z = latent.to(device=self.device, dtype=torch.bfloat16).contiguous()
        F = z.shape[0]
        assert z.shape[1:] == (16, self.lat_h, self.lat_w)
        out = torch.empty(1 + 4 * (F - 1), 3, 8 * self.lat_h, 8 * self.lat_w, dtype=torch.float32, device=self.device)
        ws = self._workspace(0)
        m, inv = self._scales(mean, std)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mmpl_vae_decode(self._h, _lib.ptr(z), F, m, inv, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "mmpl_vae_decode")
        return out

    def encode(self, pixel: torch.Tensor, mean, std) -> torch.Tensor:
        """pixel [3, T = 1 + 4k, 8h, 8w] in [-1, 1] -> float32 [1 + k, 16, h, w] (normalised mu)."""
        x = pixel.to(device=self.device, dtype=torch.bfloat16).contiguous()
        T = x.shape[1]
        assert x.shape[0] == 3 and x.shape[2:] == (8 * self.lat_h, 8 * self.lat_w) and (T - 1) % 4 == 0
        out = torch.empty(1 + (T - 1) // 4, 16, self.lat_h, self.lat_w, dtype=torch.float32, device=self.device)
        ws = self._workspace(1)
        m, inv = self._scales(mean, std)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mmpl_vae_encode(self._h, _lib.ptr(x), T, m, inv, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "mmpl_vae_encode")
        return out
