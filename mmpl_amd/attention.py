"""`attention()` seam (MMPL_t2v/wan/modules/attention.py:139-185) on the HIP flash-attention kernel.

Same signature and semantics for the arguments the FPS path uses (no mask, no dropout, softmax_scale=None ->
1/sqrt(D), bf16); arguments the HIP kernel does not implement raise instead of silently falling back.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib

__all__ = ["attention", "flash_attention"]


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0.0, softmax_scale=None, q_scale=None, causal=False,
              window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, fa_version=None):
    """q: [B, Lq, N, 128]; k, v: [B, Lk, N, 128] -> [B, Lq, N, 128] (bf16)."""
    if q_lens is not None or k_lens is not None or causal or dropout_p != 0.0 or tuple(window_size) != (-1, -1):
        raise NotImplementedError("mmpl_amd.attention: only the dense, unmasked form used by the FPS inference path is implemented")
    if q.shape[-1] != 128 or not q.is_cuda:
        raise NotImplementedError("mmpl_amd.attention: head_dim must be 128 and tensors must live on the GPU")
    lib = _lib.load()
    B, Lq, N, D = q.shape
    Lk = k.shape[1]
    q = q.to(torch.bfloat16)
    if q_scale is not None:
        q = q * q_scale
    q, k, v = q.contiguous(), k.to(torch.bfloat16).contiguous(), v.to(torch.bfloat16).contiguous()
    out = torch.empty_like(q)
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    for b in range(B):
        kp = (C.c_void_p * 1)(k[b].data_ptr())
        vp = (C.c_void_p * 1)(v[b].data_ptr())
        _lib.check(lib.mmpl_attn_fwd(_lib.ptr(q[b]), N * D, _lib.ptr(out[b]), N * D, kp, vp, N * D, N * D, 1, Lk, Lq, N,
                                     float(scale), _lib.stream_ptr()), "mmpl_attn_fwd")
    return out


flash_attention = attention
