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


class CLIPVisionTower:
    """The vision tower Wan-I2V conditions on: ``CLIPModel.visual`` (MMPL_t2v/wan/modules/clip.py:527-542) =
    ``VisionTransformer.forward(x, use_31_block=True)`` (clip.py:209-327) after a bicubic resize and the CLIP normalisation.
    ``load_state_dict`` takes the reference's ``model.visual`` keys (``patch_embedding.weight``, ``cls_embedding``,
    ``pos_embedding``, ``pre_norm.*``, ``transformer.N.{norm1,attn.to_qkv,attn.proj,norm2,mlp.0,mlp.2}.*``; ``post_norm`` and
    ``head`` are not used on this path).  Heads of 80 are padded to 128 columns at load time so that the DiT's attention
    kernel serves them (zero rows in to_qkv, zero columns in proj: bit-neutral)."""
    MEAN = (0.48145466, 0.4578275, 0.40821073)
    STD = (0.26862954, 0.26130258, 0.27577711)

    def __init__(self, image_size=224, patch_size=14, dim=1280, mlp_ratio=4, num_heads=16, num_layers=32, norm_eps=1e-5, device="cuda:0"):
        assert image_size % patch_size == 0 and dim % num_heads == 0 and dim // num_heads <= 128
        self.image_size, self.patch_size, self.dim, self.mlp_dim = image_size, patch_size, dim, int(dim * mlp_ratio)
        self.heads, self.head_dim, self.num_layers, self.eps = num_heads, dim // num_heads, num_layers, norm_eps
        self.n_patch = (image_size // patch_size) ** 2
        self.pk = (3 * patch_size * patch_size + 63) // 64 * 64
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._gw = self._lw = None

    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        dev, d, H, hd = self.device, self.dim, self.heads, self.head_dim

        def g(k):
            return sd[prefix + k].to(device=dev, dtype=BF)

        pw = g("patch_embedding.weight").reshape(d, -1)
        gw = [torch.nn.functional.pad(pw, (0, self.pk - pw.shape[1])).contiguous(), g("cls_embedding").reshape(d).contiguous(),
              g("pos_embedding").reshape(self.n_patch + 1, d).contiguous(), g("pre_norm.weight").contiguous(), g("pre_norm.bias").contiguous()]
        lw = []
        for i in range(self.num_layers - 1):                  # use_31_block: the last block is never run (clip.py:319-321)
            p = f"transformer.{i}."
            qkv_w = torch.zeros(3, H, 128, d, dtype=BF, device=dev)
            qkv_w[:, :, :hd] = g(p + "attn.to_qkv.weight").view(3, H, hd, d)
            qkv_b = torch.zeros(3, H, 128, dtype=BF, device=dev)
            qkv_b[:, :, :hd] = g(p + "attn.to_qkv.bias").view(3, H, hd)
            proj_w = torch.zeros(d, H, 128, dtype=BF, device=dev)
            proj_w[:, :, :hd] = g(p + "attn.proj.weight").view(d, H, hd)
            lw += [g(p + "norm1.weight").contiguous(), g(p + "norm1.bias").contiguous(), qkv_w.view(3 * H * 128, d), qkv_b.view(-1),
                   proj_w.view(d, H * 128), g(p + "attn.proj.bias").contiguous(), g(p + "norm2.weight").contiguous(),
                   g(p + "norm2.bias").contiguous(), g(p + "mlp.0.weight").contiguous(), g(p + "mlp.0.bias").contiguous(),
                   g(p + "mlp.2.weight").contiguous(), g(p + "mlp.2.bias").contiguous()]
        self._gw, self._lw = gw, lw

    def preprocess(self, videos) -> torch.Tensor:
        """clip.py:529-538: list of [3, T, H, W] in [-1, 1] -> [sum T, 3, S, S], bicubic-resized and CLIP-normalised (fp32)."""
        size = (self.image_size,) * 2
        x = torch.cat([torch.nn.functional.interpolate(u.transpose(0, 1).float(), size=size, mode="bicubic", align_corners=False) for u in videos])
        x = x * 0.5 + 0.5
        mean = torch.tensor(self.MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        std = torch.tensor(self.STD, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        return (x - mean) / std

    def forward_pixels(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, 3, S, S] (already normalised) -> tokens [B, n_patch + 1, dim] after num_layers - 1 blocks."""
        if self._gw is None:
            raise RuntimeError("CLIPVisionTower: weights not loaded")
        lib, P = self._lib, self.patch_size
        x = x.to(device=self.device, dtype=BF)
        B = x.shape[0]
        # im2col of the stride-P PxP convolution, (c, ky, kx) order like Conv2d's weight, K padded with zeros
        cols = torch.nn.functional.unfold(x.float(), kernel_size=P, stride=P).transpose(1, 2).to(BF)          # [B, n_patch, 3 P P]
        cols = torch.nn.functional.pad(cols, (0, self.pk - cols.shape[2])).contiguous()
        n = self.n_patch + 1
        out = torch.empty(B, n, self.dim, dtype=BF, device=self.device)
        ws = torch.empty(lib.mmpl_clip_visual_workspace_bytes(n, self.dim, self.mlp_dim, self.heads), dtype=torch.uint8, device=self.device)
        gw = (C.c_void_p * 5)(*[t.data_ptr() for t in self._gw])
        lw = (C.c_void_p * len(self._lw))(*[t.data_ptr() for t in self._lw])
        with torch.cuda.device(self.device):
            for b in range(B):
                _lib.check(lib.mmpl_clip_visual(_lib.ptr(cols[b]), self.n_patch, self.pk, self.dim, self.mlp_dim, self.heads, self.head_dim,
                                                self.num_layers - 1, gw, lw, self.eps, _lib.ptr(out[b]), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr()), "mmpl_clip_visual")
        return out

    def visual(self, videos) -> torch.Tensor:
        """CLIPModel.visual (clip.py:527-542)."""
        return self.forward_pixels(self.preprocess([v.to(self.device) for v in videos]))
