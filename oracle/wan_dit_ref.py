"""Oracle: CausalFPSWanModel inference forward, restated with plain PyTorch CPU ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows, op for op and rounding point for
rounding point, the reference's bf16 inference path:

  * ``CausalFPSWanModel._forward_inference``   MMPL_t2v/wan/modules/causal_fps_model.py:708-837
  * ``CausalWanAttentionBlock.forward``        causal_fps_model.py:312-364
  * ``CausalWanSelfAttention.forward`` (cache) causal_fps_model.py:192-269
  * ``causal_fps_rope_apply``                  causal_fps_model.py:27-55
  * ``WanT2VCrossAttention.forward``           MMPL_t2v/wan/modules/model.py:161-194
  * ``WanRMSNorm`` / ``WanLayerNorm``          model.py:70-99
  * ``sinusoidal_embedding_1d`` / ``rope_params`` model.py:15-36
  * ``attention`` (SDPA fallback)              MMPL_t2v/wan/modules/attention.py:170-185
  * ``CausalHead.forward`` / ``unpatchify``    causal_fps_model.py:384-395, 1007-1030

Differences from the reference are *interface only*: the frame geometry is a parameter
(the reference hard-codes 1560 tokens/frame), and the KV-slot write / visibility rule of
causal_fps_model.py:209-264 is passed in explicitly as ``write_slots`` / ``visible_slots``
(computed by oracle/stage_ref.py the way the reference computes them).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


@dataclass
class DitCfg:
    dim: int
    ffn_dim: int
    num_heads: int
    num_layers: int
    text_dim: int = 4096
    freq_dim: int = 256
    in_dim: int = 16
    out_dim: int = 16
    text_len: int = 512
    eps: float = 1e-6

    @property
    def head_dim(self) -> int:
        return self.dim // self.num_heads


# --------------------------------------------------------------------------- embeddings
def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """model.py:15-25 (float64)."""
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_params(max_seq_len: int, dim: int, theta: float = 10000) -> torch.Tensor:
    """model.py:29-36 -> complex128 [max_seq_len, dim/2]."""
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_table(head_dim: int) -> torch.Tensor:
    """causal_fps_model.py:510-516: concat of (t, h, w) tables -> complex128 [1024, head_dim/2]."""
    d = head_dim
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                      rope_params(1024, 2 * (d // 6))], dim=1)


def fps_rope_apply(x: torch.Tensor, frame_ids: Sequence[int], gh: int, gw: int, freqs: torch.Tensor) -> torch.Tensor:
    """causal_fps_model.py:27-55.  x: [1, L, N, D]; temporal index = absolute frame id list."""
    n, c = x.size(2), x.size(3) // 2
    f = len(frame_ids)
    fr = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    seq_len = f * gh * gw
    x_i = torch.view_as_complex(x[0, :seq_len].to(torch.float32).reshape(seq_len, n, -1, 2))
    freqs_i = torch.cat([
        fr[0][list(frame_ids)].view(f, 1, 1, -1).expand(f, gh, gw, -1),
        fr[1][:gh].view(1, gh, 1, -1).expand(f, gh, gw, -1),
        fr[2][:gw].view(1, 1, gw, -1).expand(f, gh, gw, -1)], dim=-1).reshape(seq_len, 1, -1)
    x_i = torch.view_as_real(x_i * freqs_i).flatten(2)
    return x_i.unsqueeze(0).type_as(x)


# --------------------------------------------------------------------------- norms
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """model.py:78-86: fp32 norm over the FULL last dim, cast back, then * weight."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)).type_as(x) * w


def layer_norm(x: torch.Tensor, eps: float, w: Optional[torch.Tensor] = None, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """model.py:89-99."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps).type_as(x)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """attention.py:170-185 (the reference's own non-flash path): [B,L,N,D] in/out, bf16."""
    q = q.transpose(1, 2).to(torch.bfloat16)
    k = k.transpose(1, 2).to(torch.bfloat16)
    v = v.transpose(1, 2).to(torch.bfloat16)
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=None, is_causal=False, dropout_p=0.0)
    return out.transpose(1, 2).contiguous()


def sdpa_fp32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """fp32 restatement of the same attention (for two-sided tolerance checks)."""
    qf, kf, vf = (u.transpose(1, 2).float() for u in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).contiguous()


# --------------------------------------------------------------------------- model pieces
def _lin(x, p, name):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def embed_context(p: Dict[str, torch.Tensor], cfg: DitCfg, context: torch.Tensor) -> torch.Tensor:
    """causal_fps_model.py:780-786: zero-pad to text_len, Linear-GELU(tanh)-Linear.  -> [1, text_len, dim]."""
    u = context
    u = torch.cat([u, u.new_zeros(cfg.text_len - u.size(0), u.size(1))]).unsqueeze(0)
    return _lin(F.gelu(_lin(u, p, "text_embedding.0"), approximate="tanh"), p, "text_embedding.2")


def time_embed(p, cfg: DitCfg, t: torch.Tensor, like: torch.Tensor):
    """causal_fps_model.py:773-776.  t: [1, nF] -> e [nF, dim], e0 [1, nF, 6, dim]."""
    e = sinusoidal_embedding_1d(cfg.freq_dim, t.flatten()).type_as(like)
    e = _lin(F.silu(_lin(e, p, "time_embedding.0")), p, "time_embedding.2")
    e0 = _lin(F.silu(e), p, "time_projection.1").unflatten(1, (6, cfg.dim)).unflatten(dim=0, sizes=t.shape)
    return e, e0


def cross_kv(p, cfg: DitCfg, layer: int, context_emb: torch.Tensor):
    """model.py:175-180: cached text K (RMS-normed) / V, [1, text_len, N, D]."""
    pre = f"blocks.{layer}.cross_attn."
    n, d = cfg.num_heads, cfg.head_dim
    k = rms_norm(_lin(context_emb, p, pre + "k"), p[pre + "norm_k.weight"], cfg.eps).view(1, -1, n, d)
    v = _lin(context_emb, p, pre + "v").view(1, -1, n, d)
    return k, v


def self_attention(p, cfg: DitCfg, layer: int, x: torch.Tensor, kv: Dict[str, torch.Tensor], frame_ids, write_slots,
                   visible_slots, S: int, gh: int, gw: int, freqs, attn_fn=sdpa) -> torch.Tensor:
    """causal_fps_model.py:104-115 + 192-269 with the slot rule made explicit.

    kv["k"], kv["v"]: [1, n_slots*S, N, D].  write_slots[i] >= 0: frame i's roped K and V are
    copied into that slot *before* attending (lines 211-217 / 229-241).  write_slots all -1:
    the stage's own K/V are concatenated after the gathered cache instead (lines 254-264).
    visible_slots: cache slots gathered for attention (lines 219-227).
    """
    pre = f"blocks.{layer}.self_attn."
    b, s, n, d = 1, x.shape[1], cfg.num_heads, cfg.head_dim
    q = rms_norm(_lin(x, p, pre + "q"), p[pre + "norm_q.weight"], cfg.eps).view(b, s, n, d)
    k = rms_norm(_lin(x, p, pre + "k"), p[pre + "norm_k.weight"], cfg.eps).view(b, s, n, d)
    v = _lin(x, p, pre + "v").view(b, s, n, d)
    rq = fps_rope_apply(q, frame_ids, gh, gw, freqs).type_as(v)
    rk = fps_rope_apply(k, frame_ids, gh, gw, freqs).type_as(v)
    writes = [w for w in write_slots if w >= 0]
    if writes:
        assert len(writes) == len(write_slots)
        for i, slot in enumerate(write_slots):
            kv["k"][:, slot * S:(slot + 1) * S].copy_(rk[:, i * S:(i + 1) * S])
            kv["v"][:, slot * S:(slot + 1) * S].copy_(v[:, i * S:(i + 1) * S])
    idx = [j for slot in visible_slots for j in range(slot * S, (slot + 1) * S)]
    kk, vv = kv["k"][:, idx], kv["v"][:, idx]
    if not writes:
        kk, vv = torch.cat([kk, rk], dim=1), torch.cat([vv, v], dim=1)
    out = attn_fn(rq, kk, vv)
    return _lin(out.flatten(2), p, pre + "o")


def image_kv(p, cfg: DitCfg, layer: int, ctx_img: torch.Tensor):
    """WanI2VCrossAttention (model.py:251-252): K (RMS-normed) / V of the projected CLIP tokens, [1, 257, N, D]."""
    pre = f"blocks.{layer}.cross_attn."
    n, d = cfg.num_heads, cfg.head_dim
    k = rms_norm(_lin(ctx_img, p, pre + "k_img"), p[pre + "norm_k_img.weight"], cfg.eps).view(1, -1, n, d)
    v = _lin(ctx_img, p, pre + "v_img").view(1, -1, n, d)
    return k, v


def cross_attention(p, cfg: DitCfg, layer: int, x: torch.Tensor, ck: torch.Tensor, cv: torch.Tensor, attn_fn=sdpa, img=None):
    """model.py:161-194 with the K/V cache already initialised; img = (k_img, v_img): the Wan-I2V form (model.py:238-266),
    whose image attention output is added (in bf16) before the o-projection."""
    pre = f"blocks.{layer}.cross_attn."
    q = rms_norm(_lin(x, p, pre + "q"), p[pre + "norm_q.weight"], cfg.eps).view(1, -1, cfg.num_heads, cfg.head_dim)
    out = attn_fn(q, ck, cv).flatten(2)
    if img is not None:
        out = out + attn_fn(q, img[0], img[1]).flatten(2)
    return _lin(out, p, pre + "o")


def block_forward(p, cfg: DitCfg, layer: int, x, e0, kv, ck, cv, frame_ids, write_slots, visible_slots, S, gh, gw,
                  freqs, attn_fn=sdpa, img=None):
    """causal_fps_model.py:335-364."""
    nF = e0.shape[1]
    pre = f"blocks.{layer}."
    e = (p[pre + "modulation"].unsqueeze(1) + e0).chunk(6, dim=2)
    y = self_attention(
        p, cfg, layer,
        (layer_norm(x, cfg.eps).unflatten(dim=1, sizes=(nF, S)) * (1 + e[1]) + e[0]).flatten(1, 2),
        kv, frame_ids, write_slots, visible_slots, S, gh, gw, freqs, attn_fn)
    x = x + (y.unflatten(dim=1, sizes=(nF, S)) * e[2]).flatten(1, 2)
    x = x + cross_attention(p, cfg, layer, layer_norm(x, cfg.eps, p[pre + "norm3.weight"], p[pre + "norm3.bias"]),
                            ck, cv, attn_fn, img)
    h = (layer_norm(x, cfg.eps).unflatten(dim=1, sizes=(nF, S)) * (1 + e[4]) + e[3]).flatten(1, 2)
    y = _lin(F.gelu(_lin(h, p, pre + "ffn.0"), approximate="tanh"), p, pre + "ffn.2")
    x = x + (y.unflatten(dim=1, sizes=(nF, S)) * e[5]).flatten(1, 2)
    return x


def head_forward(p, cfg: DitCfg, x, e, nF, S):
    """causal_fps_model.py:384-395 (e: [1, nF, 1, dim])."""
    em = (p["head.modulation"].unsqueeze(1) + e).chunk(2, dim=2)
    return _lin(layer_norm(x, cfg.eps).unflatten(dim=1, sizes=(nF, S)) * (1 + em[1]) + em[0], p, "head.head")


def unpatchify(x: torch.Tensor, cfg: DitCfg, nF: int, gh: int, gw: int) -> torch.Tensor:
    """causal_fps_model.py:1007-1030.  x: [nF*gh*gw, 4*out_dim] -> [out_dim, nF, 2gh, 2gw]."""
    c = cfg.out_dim
    u = x[:nF * gh * gw].view(nF, gh, gw, 1, 2, 2, c)
    u = torch.einsum("fhwpqrc->cfphqwr", u)
    return u.reshape(c, nF, gh * 2, gw * 2)


def new_kv_cache(cfg: DitCfg, n_slots: int, S: int, dtype=torch.bfloat16):
    """casual_fps_inference.py:453-480 (layout [1, n_slots*S, N, D] per layer)."""
    return [{"k": torch.zeros(1, n_slots * S, cfg.num_heads, cfg.head_dim, dtype=dtype),
             "v": torch.zeros(1, n_slots * S, cfg.num_heads, cfg.head_dim, dtype=dtype)} for _ in range(cfg.num_layers)]


def dit_forward(p: Dict[str, torch.Tensor], cfg: DitCfg, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor,
                kv_cache: List[Dict[str, torch.Tensor]], cross_cache: List[Optional[tuple]], frame_ids: Sequence[int],
                write_slots: Sequence[int], visible_slots: Sequence[int], attn_fn=sdpa,
                return_hidden: bool = False, clip_fea: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One inference forward (causal_fps_model.py:708-837).

    x: [in_dim, nF, h, w]; t: [1, nF] float32; context: [L<=text_len, text_dim];
    cross_cache[layer] is None (-> filled, model.py:175-180) or (k, v).  Returns [out_dim, nF, h, w].
    clip_fea [257, 1280]: Wan-I2V model type (model.py:672-712) -- x already carries the conditioning video on its channel
    axis (in_dim 36), the CLIP tokens go through img_emb (MLPProj) and every block's cross-attention also attends to them.
    """
    nF, gh, gw = x.shape[1], x.shape[2] // 2, x.shape[3] // 2
    S = gh * gw
    freqs = rope_table(cfg.head_dim).to(x.device)
    h = F.conv3d(x.unsqueeze(0), p["patch_embedding.weight"], p["patch_embedding.bias"], stride=(1, 2, 2))
    h = h.flatten(2).transpose(1, 2)                                   # [1, nF*S, dim]
    e, e0 = time_embed(p, cfg, t, h)
    ctx = embed_context(p, cfg, context)
    ctx_img = None
    if clip_fea is not None:
        from .i2v_ref import mlp_proj
        ctx_img = mlp_proj({k[len("img_emb."):]: v for k, v in p.items() if k.startswith("img_emb.")}, clip_fea.unsqueeze(0))
    for layer in range(cfg.num_layers):
        if cross_cache[layer] is None:
            cross_cache[layer] = cross_kv(p, cfg, layer, ctx)
        ck, cv = cross_cache[layer]
        h = block_forward(p, cfg, layer, h, e0, kv_cache[layer], ck, cv, frame_ids, write_slots, visible_slots,
                          S, gh, gw, freqs, attn_fn, image_kv(p, cfg, layer, ctx_img) if ctx_img is not None else None)
    if return_hidden:
        return h
    y = head_forward(p, cfg, h, e.unflatten(dim=0, sizes=t.shape).unsqueeze(2), nF, S)
    return unpatchify(y.flatten(0, 2), cfg, nF, gh, gw)
