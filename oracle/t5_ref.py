"""Oracle: umT5 encoder (the text encoder in front of the denoising path), restated functionally.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows MMPL_t2v/wan/modules/t5.py: T5LayerNorm :52-67, T5Attention
:70-121 (no score scaling, additive position bias, key mask filled with finfo.min, fp32 softmax), T5FeedForward
:124-145 with the hand-written GELU :45-49, T5SelfAttention :148-179, T5RelativeEmbedding :225-264 (bidirectional
buckets), T5Encoder :267-312 (per-layer position embedding: shared_pos=False for umt5-xxl, :456-469), and the
wrapper's pad zeroing MMPL_t2v/utils/wan_wrapper.py:38-51.  Weights: plain state_dict with the reference's keys.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def t5_layer_norm(x, w, eps=1e-6):
    x = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        x = x.type_as(w)
    return w * x


def gelu_t5(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def relative_position_bucket(rel_pos: torch.Tensor, num_buckets: int, max_dist: int = 128) -> torch.Tensor:
    """t5.py:240-264, bidirectional."""
    nb = num_buckets // 2
    rel_buckets = (rel_pos > 0).long() * nb
    rel_pos = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(rel_pos.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return rel_buckets + torch.where(rel_pos < max_exact, rel_pos, large)


def position_bias(emb_w: torch.Tensor, lq: int, lk: int, num_buckets: int) -> torch.Tensor:
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    return F.embedding(relative_position_bucket(rel, num_buckets), emb_w).permute(2, 0, 1).unsqueeze(0).contiguous()


def attention(p, pre, x, mask, pos_bias, n):
    b, c = x.size(0), p[pre + "q.weight"].shape[0] // n
    q = F.linear(x, p[pre + "q.weight"]).view(b, -1, n, c)
    k = F.linear(x, p[pre + "k.weight"]).view(b, -1, n, c)
    v = F.linear(x, p[pre + "v.weight"]).view(b, -1, n, c)
    attn_bias = x.new_zeros(b, n, q.size(1), k.size(1))
    attn_bias += pos_bias
    attn_bias.masked_fill_(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + attn_bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    x = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, -1, n * c)
    return F.linear(x, p[pre + "o.weight"])


def encoder_forward(p: Dict[str, torch.Tensor], ids: torch.Tensor, mask: torch.Tensor, num_heads: int, num_buckets: int,
                    num_layers: int) -> torch.Tensor:
    """ids, mask: [B, L] -> [B, L, dim]."""
    x = F.embedding(ids, p["token_embedding.weight"])
    L = x.size(1)
    for i in range(num_layers):
        pre = f"blocks.{i}."
        e = position_bias(p[pre + "pos_embedding.embedding.weight"], L, L, num_buckets)
        x = x + attention(p, pre + "attn.", t5_layer_norm(x, p[pre + "norm1.weight"]), mask, e, num_heads)
        h = t5_layer_norm(x, p[pre + "norm2.weight"])
        h = F.linear(h, p[pre + "ffn.fc1.weight"]) * gelu_t5(F.linear(h, p[pre + "ffn.gate.0.weight"]))
        x = x + F.linear(h, p[pre + "ffn.fc2.weight"])
    return t5_layer_norm(x, p["norm.weight"])


def text_encoder_forward(p, ids, mask, num_heads, num_buckets, num_layers):
    """WanTextEncoder.forward (wan_wrapper.py:38-51): encoder output with the padding rows zeroed."""
    ctx = encoder_forward(p, ids, mask, num_heads, num_buckets, num_layers)
    for u, v in zip(ctx, mask.gt(0).sum(dim=1).long()):
        u[v:] = 0.0
    return ctx
