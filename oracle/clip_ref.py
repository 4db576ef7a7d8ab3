"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the reference's CLIP vision tower as Wan-I2V uses it:
``VisionTransformer.forward(x, use_31_block=True)`` (MMPL_t2v/wan/modules/clip.py:209-327) with its ``AttentionBlock``
(clip.py:120-155, pre-norm, GELU) and ``SelfAttention`` (clip.py:52-86), functional, on a state_dict.  Pinned bit-exactly
against the imported reference module by tests/golden/make_golden_clip.py."""
from typing import Dict

import torch
import torch.nn.functional as F

from .wan_dit_ref import sdpa


def _ln(x, w, b, eps):
    """clip.py:45-48: LayerNorm in fp32, result cast back"""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).type_as(x)


def clip_visual(p: Dict[str, torch.Tensor], x: torch.Tensor, num_heads: int, num_layers: int, patch_size: int, eps: float = 1e-5) -> torch.Tensor:
    """x: [B, 3, S, S] normalised pixels -> [B, (S/patch)^2 + 1, dim] after num_layers - 1 blocks (use_31_block)."""
    b = x.size(0)
    x = F.conv2d(x, p["patch_embedding.weight"], None, stride=patch_size).flatten(2).permute(0, 2, 1)
    x = torch.cat([p["cls_embedding"].expand(b, -1, -1), x], dim=1)
    x = x + p["pos_embedding"]
    x = _ln(x, p["pre_norm.weight"], p["pre_norm.bias"], eps)
    for i in range(num_layers - 1):
        pre = f"transformer.{i}."
        h = _ln(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
        s, c = h.shape[1], h.shape[2]
        q, k, v = F.linear(h, p[pre + "attn.to_qkv.weight"], p[pre + "attn.to_qkv.bias"]).view(b, s, 3, num_heads, c // num_heads).unbind(2)
        a = sdpa(q, k, v).reshape(b, s, c)
        x = x + F.linear(a, p[pre + "attn.proj.weight"], p[pre + "attn.proj.bias"])
        h = _ln(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
        h = F.linear(F.gelu(F.linear(h, p[pre + "mlp.0.weight"], p[pre + "mlp.0.bias"])), p[pre + "mlp.2.weight"], p[pre + "mlp.2.bias"])
        x = x + h
    return x
