"""Wan-I2V image cross-attention on the HIP kernels: drop-in counterparts of the reference's
``MLPProj`` (MMPL_t2v/wan/modules/model.py:469-481) and ``WanI2VCrossAttention`` (model.py:224-266), taking their
state_dicts as they are.  The image K/V (like the text K/V) depend only on the prompt/image, so they are computed once
(`prepare`) and the per-forward work is q-projection, two attentions, their sum and the o-projection."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib

BF = torch.bfloat16


class MLPProj:
    KEYS = ["proj.0.weight", "proj.0.bias", "proj.1.weight", "proj.1.bias", "proj.3.weight", "proj.3.bias", "proj.4.weight", "proj.4.bias"]

    def __init__(self, in_dim: int, out_dim: int, device="cuda:0"):
        self.in_dim, self.out_dim, self.device = in_dim, out_dim, torch.device(device)
        self._lib = _lib.load()
        self._w = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        self._w = [sd[prefix + k].to(device=self.device, dtype=BF).contiguous() for k in self.KEYS]

    def __call__(self, image_embeds: torch.Tensor) -> torch.Tensor:
        """image_embeds: [B, 257, in_dim] -> clip_extra_context_tokens [B, 257, out_dim] (model.py:479-481)."""
        if self._w is None:
            raise RuntimeError("MLPProj: weights not loaded")
        x = image_embeds.to(device=self.device, dtype=BF).contiguous()
        B, n, _ = x.shape
        out = torch.empty(B, n, self.out_dim, dtype=BF, device=self.device)
        ws = torch.empty(self._lib.mmpl_i2v_img_proj_workspace_bytes(n, self.in_dim, self.out_dim), dtype=torch.uint8, device=self.device)
        arr = (C.c_void_p * 8)(*[t.data_ptr() for t in self._w])
        for b in range(B):
            _lib.check(self._lib.mmpl_i2v_img_proj(_lib.ptr(x[b]), n, self.in_dim, self.out_dim, arr, _lib.ptr(out[b]), _lib.ptr(ws),
                                                   ws.numel(), _lib.stream_ptr()), "mmpl_i2v_img_proj")
        return out


class WanI2VCrossAttention:
    def __init__(self, dim: int, num_heads: int, eps: float = 1e-6, device="cuda:0"):
        assert dim // num_heads == 128, "head_dim 128 (both Wan2.1 models)"
        self.dim, self.num_heads, self.eps, self.device = dim, num_heads, eps, torch.device(device)
        self._lib = _lib.load()
        self._p: Optional[Dict[str, torch.Tensor]] = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        names = [f"{m}.{t}" for m in ("q", "k", "v", "o", "k_img", "v_img") for t in ("weight", "bias")] + \
                ["norm_q.weight", "norm_k.weight", "norm_k_img.weight"]
        self._p = {k: sd[prefix + k].to(device=self.device, dtype=BF).contiguous() for k in names}

    def prepare(self, context: torch.Tensor):
        """context: [257 + L_txt, dim] (image tokens first, model.py:245-246) -> (k_txt, v_txt, k_img, v_img)."""
        p, d = self._p, self.dim
        ctx = context.to(device=self.device, dtype=BF).contiguous()
        img, txt = ctx[:257].contiguous(), ctx[257:].contiguous()
        k_img, v_img = torch.empty_like(img), torch.empty_like(img)
        _lib.check(self._lib.mmpl_i2v_img_kv(_lib.ptr(img), img.shape[0], d, _lib.ptr(p["k_img.weight"]), _lib.ptr(p["k_img.bias"]),
                                             _lib.ptr(p["v_img.weight"]), _lib.ptr(p["v_img.bias"]), _lib.ptr(p["norm_k_img.weight"]),
                                             self.eps, _lib.ptr(k_img), _lib.ptr(v_img), _lib.stream_ptr()), "mmpl_i2v_img_kv")
        k_txt, v_txt = torch.empty_like(txt), torch.empty_like(txt)     # the same op computes norm_k(k(context)), v(context)
        _lib.check(self._lib.mmpl_i2v_img_kv(_lib.ptr(txt), txt.shape[0], d, _lib.ptr(p["k.weight"]), _lib.ptr(p["k.bias"]),
                                             _lib.ptr(p["v.weight"]), _lib.ptr(p["v.bias"]), _lib.ptr(p["norm_k.weight"]),
                                             self.eps, _lib.ptr(k_txt), _lib.ptr(v_txt), _lib.stream_ptr()), "mmpl_i2v_img_kv")
        return k_txt, v_txt, k_img, v_img

    def forward(self, x: torch.Tensor, context: torch.Tensor, context_lens=None, kv=None) -> torch.Tensor:
        """x: [1, L1, dim]; context: [1, 257 + L2, dim] -> [1, L1, dim] (model.py:238-266; context_lens is None on this
        path -- the reference's callers pass full-length, zero-padded text)."""
        if self._p is None:
            raise RuntimeError("WanI2VCrossAttention: weights not loaded")
        assert x.shape[0] == 1 and context_lens is None
        p, d = self._p, self.dim
        k_txt, v_txt, k_img, v_img = kv if kv is not None else self.prepare(context[0])
        xx = x[0].to(device=self.device, dtype=BF).contiguous()
        Lq = xx.shape[0]
        out = torch.empty_like(xx)
        ws = torch.empty(self._lib.mmpl_i2v_cross_attn_workspace_bytes(Lq, d), dtype=torch.uint8, device=self.device)
        _lib.check(self._lib.mmpl_i2v_cross_attn(_lib.ptr(xx), Lq, d, _lib.ptr(p["q.weight"]), _lib.ptr(p["q.bias"]),
                                                 _lib.ptr(p["norm_q.weight"]), self.eps, _lib.ptr(k_txt), _lib.ptr(v_txt), k_txt.shape[0],
                                                 _lib.ptr(k_img), _lib.ptr(v_img), k_img.shape[0], _lib.ptr(p["o.weight"]),
                                                 _lib.ptr(p["o.bias"]), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                   "mmpl_i2v_cross_attn")
        return out.unsqueeze(0)

    __call__ = forward
