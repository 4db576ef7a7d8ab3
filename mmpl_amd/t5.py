"""Host-side handle of the HIP umT5 encoder (libmmpl_hip.so: mmpl_t5_*), the engine behind ``WanTextEncoder``.

Takes the reference's ``models_t5_umt5-xxl-enc-bf16.pth`` state_dict as is (MMPL_t2v/wan/modules/t5.py:267-312).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List

import torch

from . import _lib


def relative_position_buckets(text_len: int, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """bucket of rel = j - i for rel in [-(L-1), L-1] (t5.py:240-264, bidirectional), int32 [2L-1]."""
    rel = torch.arange(-(text_len - 1), text_len)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return (out + torch.where(rel < max_exact, rel, large)).to(torch.int32)


class T5Engine:
    def __init__(self, cfg: dict, text_len: int = 512, device="cuda:0"):
        self.cfg, self.text_len, self.device = dict(cfg), text_len, torch.device(device)
        self._lib = _lib.load()
        self._c = _lib.MmplT5Config(vocab=cfg["vocab"], dim=cfg["dim"], dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"],
                                    num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_buckets=cfg["num_buckets"],
                                    text_len=text_len, eps=1e-6)
        h = C.c_void_p()
        with torch.cuda.device(self.device):          # the handle, its stream and every launch belong to THIS device
            _lib.check(self._lib.mmpl_t5_create(C.byref(self._c), C.byref(h)), "mmpl_t5_create")
        self._h = h
        self._bucket = relative_position_buckets(text_len, cfg["num_buckets"]).to(self.device)
        self._weights: List[torch.Tensor] = []
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mmpl_t5_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        def g(k):
            return sd[prefix + k].to(device=self.device, dtype=torch.bfloat16).contiguous()
        w = [g("token_embedding.weight"), g("norm.weight")]
        for i in range(self.cfg["num_layers"]):
            p = f"blocks.{i}."
            w += [g(p + "norm1.weight"),
                  torch.cat([g(p + "attn.q.weight"), g(p + "attn.k.weight"), g(p + "attn.v.weight")]).contiguous(),
                  g(p + "attn.o.weight"), g(p + "pos_embedding.embedding.weight"), g(p + "norm2.weight"),
                  g(p + "ffn.gate.0.weight"), g(p + "ffn.fc1.weight"), g(p + "ffn.fc2.weight")]
        n = self._lib.mmpl_t5_num_weights(C.byref(self._c))
        assert len(w) == n
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in w])
        _lib.check(self._lib.mmpl_t5_bind_weights(self._h, arr, n), "mmpl_t5_bind_weights")
        self._weights = w

    def encode(self, ids: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """ids, mask: [B, text_len] integer -> [B, text_len, dim] bf16 with padding rows zeroed (WanTextEncoder.forward)."""
        B, L = ids.shape
        assert L == self.text_len
        if self._ws is None:
            self._ws = torch.empty(self._lib.mmpl_t5_workspace_bytes(self._h), dtype=torch.uint8, device=self.device)
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        mask = mask.to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty(B, L, self.cfg["dim"], dtype=torch.bfloat16, device=self.device)
        with torch.cuda.device(self.device):          # _lib.stream_ptr() = the current stream of the current device
            for b in range(B):
                _lib.check(self._lib.mmpl_t5_encode(self._h, _lib.ptr(ids[b]), _lib.ptr(mask[b]), _lib.ptr(self._bucket), _lib.ptr(out[b]),
                                                    _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "mmpl_t5_encode")
        return out
