"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the reference's Wan-I2V image conditioning:
MLPProj (MMPL_t2v/wan/modules/model.py:469-481) and WanI2VCrossAttention.forward (model.py:238-266), functional, on a
state_dict.  Pinned bit-exactly against the imported reference modules by tests/golden/make_golden_i2v.py."""
from typing import Dict

import torch
import torch.nn.functional as F

from .wan_dit_ref import rms_norm, sdpa


def mlp_proj(p: Dict[str, torch.Tensor], image_embeds: torch.Tensor) -> torch.Tensor:
    """model.py:474-481: LayerNorm -> Linear -> GELU (erf) -> Linear -> LayerNorm (torch default eps 1e-5)."""
    x = F.layer_norm(image_embeds, (image_embeds.shape[-1],), p["proj.0.weight"], p["proj.0.bias"], 1e-5)
    x = F.gelu(F.linear(x, p["proj.1.weight"], p["proj.1.bias"]))
    x = F.linear(x, p["proj.3.weight"], p["proj.3.bias"])
    return F.layer_norm(x, (x.shape[-1],), p["proj.4.weight"], p["proj.4.bias"], 1e-5)


def i2v_cross_attention(p: Dict[str, torch.Tensor], x: torch.Tensor, context: torch.Tensor, num_heads: int, eps: float = 1e-6,
                        attn_fn=sdpa) -> torch.Tensor:
    """model.py:238-266.  x: [B, L1, C]; context: [B, 257 + L2, C] (image tokens first)."""
    context_img, context = context[:, :257], context[:, 257:]
    b, n, d = x.size(0), num_heads, x.size(-1) // num_heads
    lin = lambda t, name: F.linear(t, p[name + ".weight"], p[name + ".bias"])
    q = rms_norm(lin(x, "q"), p["norm_q.weight"], eps).view(b, -1, n, d)
    k = rms_norm(lin(context, "k"), p["norm_k.weight"], eps).view(b, -1, n, d)
    v = lin(context, "v").view(b, -1, n, d)
    k_img = rms_norm(lin(context_img, "k_img"), p["norm_k_img.weight"], eps).view(b, -1, n, d)
    v_img = lin(context_img, "v_img").view(b, -1, n, d)
    img_x = attn_fn(q, k_img, v_img)
    x = attn_fn(q, k, v)
    x = x.flatten(2) + img_x.flatten(2)
    return lin(x, "o")
