"""Oracle: Wan 3D causal VAE (decoder + encoder with the per-conv temporal feature cache), restated functionally.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows MMPL_t2v/wan/modules/vae.py:
  * CausalConv3d :17-36, RMS_norm :39-54, Upsample :57-63, Resample :66-160, ResidualBlock :186-220,
    AttentionBlock :223-262, Encoder3d :265-366, Decoder3d :369-472, WanVAE_.encode/decode :517-569,
    config dim=96, z=16, dim_mult [1,2,4,4], temperal_downsample [False,True,True] :617-624
and the wrapper's scaling/clamp (MMPL_t2v/utils/wan_wrapper.py:74-113).
Weights are a plain state_dict with the reference's key names (``decoder.upsamples.3.residual.2.weight`` ...).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

CACHE_T = 2
DIM, Z = 96, 16
DIM_MULT = [1, 2, 4, 4]
T_DOWN = [False, True, True]


def causal_conv3d(x, w, b, pad, cache_x=None, stride=(1, 1, 1)):
    """vae.py:28-36.  pad = (pt, ph, pw) of the module's `padding` argument."""
    padding = [pad[2], pad[2], pad[1], pad[1], 2 * pad[0], 0]
    if cache_x is not None and padding[4] > 0:
        x = torch.cat([cache_x, x], dim=2)
        padding[4] -= cache_x.shape[2]
    x = F.pad(x, padding)
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x, gamma, channel_dim=1):
    """vae.py:51-54 (bias=False everywhere in this VAE)."""
    return F.normalize(x, dim=channel_dim) * (x.shape[channel_dim] ** 0.5) * gamma


def _cache_update(x, cache):
    """the recurring idiom vae.py:207-214: keep the last two frames of the conv INPUT."""
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and cache is not None:
        cache_x = torch.cat([cache[:, :, -1].unsqueeze(2), cache_x], dim=2)
    return cache_x


def residual_block(p, pre, x, feat, idx, in_dim, out_dim):
    """vae.py:202-220."""
    h = causal_conv3d(x, p[pre + "shortcut.weight"], p[pre + "shortcut.bias"], (0, 0, 0)) if in_dim != out_dim else x
    for norm_i, conv_i in ((0, 2), (3, 6)):
        x = F.silu(rms_norm(x, p[pre + f"residual.{norm_i}.gamma"]))
        cache_x = _cache_update(x, feat[idx[0]])
        x = causal_conv3d(x, p[pre + f"residual.{conv_i}.weight"], p[pre + f"residual.{conv_i}.bias"], (1, 1, 1), feat[idx[0]])
        feat[idx[0]] = cache_x
        idx[0] += 1
    return x + h


def attention_block(p, pre, x):
    """vae.py:240-262 (single head over h*w tokens, per frame)."""
    identity = x
    b, c, t, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    x = rms_norm(x, p[pre + "norm.gamma"])
    qkv = F.conv2d(x, p[pre + "to_qkv.weight"], p[pre + "to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    x = F.scaled_dot_product_attention(q, k, v)
    x = x.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    x = F.conv2d(x, p[pre + "proj.weight"], p[pre + "proj.bias"])
    x = x.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return x + identity


def resample_up(p, pre, x, feat, idx, mode):
    """vae.py:101-141 (upsample2d / upsample3d)."""
    b, c, t, h, w = x.shape
    if mode == "upsample3d":
        i = idx[0]
        if feat[i] is None:
            feat[i] = "Rep"
            idx[0] += 1
        else:
            cache_x = x[:, :, -CACHE_T:].clone()
            if cache_x.shape[2] < 2 and not isinstance(feat[i], str):
                cache_x = torch.cat([feat[i][:, :, -1].unsqueeze(2), cache_x], dim=2)
            if cache_x.shape[2] < 2 and isinstance(feat[i], str):
                cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
            w_, b_ = p[pre + "time_conv.weight"], p[pre + "time_conv.bias"]
            x = causal_conv3d(x, w_, b_, (1, 0, 0), None if isinstance(feat[i], str) else feat[i])
            feat[i] = cache_x
            idx[0] += 1
            x = x.reshape(b, 2, c, t, h, w)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)
    t = x.shape[2]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    x = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest").type_as(x)
    x = F.conv2d(x, p[pre + "resample.1.weight"], p[pre + "resample.1.bias"], padding=1)
    return x.reshape(b, t, c // 2, h * 2, w * 2).permute(0, 2, 1, 3, 4)


def resample_down(p, pre, x, feat, idx, mode):
    """vae.py:138-160 (downsample2d / downsample3d)."""
    b, c, t, h, w = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    x = F.conv2d(F.pad(x, (0, 1, 0, 1)), p[pre + "resample.1.weight"], p[pre + "resample.1.bias"], stride=(2, 2))
    x = x.reshape(b, t, c, h // 2, w // 2).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d":
        i = idx[0]
        if feat[i] is None:
            feat[i] = x.clone()
            idx[0] += 1
        else:
            cache_x = x[:, :, -1:].clone()
            x = F.conv3d(torch.cat([feat[i][:, :, -1:], x], 2), p[pre + "time_conv.weight"], p[pre + "time_conv.bias"],
                         stride=(2, 1, 1))
            feat[i] = cache_x
            idx[0] += 1
    return x


def _cached_conv(p, name, x, feat, idx):
    cache_x = _cache_update(x, feat[idx[0]])
    x = causal_conv3d(x, p[name + ".weight"], p[name + ".bias"], (1, 1, 1), feat[idx[0]])
    feat[idx[0]] = cache_x
    idx[0] += 1
    return x


def decoder_plan():
    """vae.py:388-416 -> list of ('res', in, out) / ('up', dim, mode) in module-index order of decoder.upsamples."""
    dims = [DIM * u for u in [DIM_MULT[-1]] + DIM_MULT[::-1]]
    t_up = T_DOWN[::-1]
    plan = []
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            in_dim = in_dim // 2
        for _ in range(3):
            plan.append(("res", in_dim, out_dim))
            in_dim = out_dim
        if i != len(DIM_MULT) - 1:
            plan.append(("up", out_dim, "upsample3d" if t_up[i] else "upsample2d"))
    return dims, plan


def encoder_plan():
    """vae.py:284-306."""
    dims = [DIM * u for u in [1] + DIM_MULT]
    plan = []
    for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(2):
            plan.append(("res", in_dim, out_dim))
            in_dim = out_dim
        if i != len(DIM_MULT) - 1:
            plan.append(("down", out_dim, "downsample3d" if T_DOWN[i] else "downsample2d"))
    return dims, plan


N_DEC_CACHE, N_ENC_CACHE = 33, 26      # CausalConv3d counts


def decoder_forward(p, x, feat):
    """vae.py:423-472."""
    idx = [0]
    dims, plan = decoder_plan()
    x = _cached_conv(p, "decoder.conv1", x, feat, idx)
    x = residual_block(p, "decoder.middle.0.", x, feat, idx, dims[0], dims[0])
    x = attention_block(p, "decoder.middle.1.", x)
    x = residual_block(p, "decoder.middle.2.", x, feat, idx, dims[0], dims[0])
    for j, item in enumerate(plan):
        pre = f"decoder.upsamples.{j}."
        if item[0] == "res":
            x = residual_block(p, pre, x, feat, idx, item[1], item[2])
        else:
            x = resample_up(p, pre, x, feat, idx, item[2])
    x = F.silu(rms_norm(x, p["decoder.head.0.gamma"]))
    return _cached_conv(p, "decoder.head.2", x, feat, idx)


def encoder_forward(p, x, feat):
    """vae.py:318-366."""
    idx = [0]
    dims, plan = encoder_plan()
    x = _cached_conv(p, "encoder.conv1", x, feat, idx)
    for j, item in enumerate(plan):
        pre = f"encoder.downsamples.{j}."
        if item[0] == "res":
            x = residual_block(p, pre, x, feat, idx, item[1], item[2])
        else:
            x = resample_down(p, pre, x, feat, idx, item[2])
    x = residual_block(p, "encoder.middle.0.", x, feat, idx, dims[-1], dims[-1])
    x = attention_block(p, "encoder.middle.1.", x)
    x = residual_block(p, "encoder.middle.2.", x, feat, idx, dims[-1], dims[-1])
    x = F.silu(rms_norm(x, p["encoder.head.0.gamma"]))
    return _cached_conv(p, "encoder.head.2", x, feat, idx)


def decode(p: Dict[str, torch.Tensor], z: torch.Tensor, mean: torch.Tensor, inv_std: torch.Tensor) -> torch.Tensor:
    """WanVAE_.decode, vae.py:545-569.  z: [1, 16, T, h, w] -> [1, 3, 1+4(T-1), 8h, 8w]."""
    z = z / inv_std.view(1, Z, 1, 1, 1) + mean.view(1, Z, 1, 1, 1)
    x = causal_conv3d(z, p["conv2.weight"], p["conv2.bias"], (0, 0, 0))
    feat: List[Optional[torch.Tensor]] = [None] * N_DEC_CACHE
    outs = [decoder_forward(p, x[:, :, i:i + 1], feat) for i in range(z.shape[2])]
    return torch.cat(outs, 2)


def encode(p: Dict[str, torch.Tensor], x: torch.Tensor, mean: torch.Tensor, inv_std: torch.Tensor) -> torch.Tensor:
    """WanVAE_.encode, vae.py:517-543.  x: [1, 3, 1+4k, H, W] -> mu [1, 16, 1+k, H/8, W/8] (normalised)."""
    feat: List[Optional[torch.Tensor]] = [None] * N_ENC_CACHE
    t = x.shape[2]
    outs = []
    for i in range(1 + (t - 1) // 4):
        sl = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder_forward(p, sl, feat))
    out = torch.cat(outs, 2)
    mu, _ = causal_conv3d(out, p["conv1.weight"], p["conv1.bias"], (0, 0, 0)).chunk(2, dim=1)
    return (mu - mean.view(1, Z, 1, 1, 1)) * inv_std.view(1, Z, 1, 1, 1)


def decode_to_pixel(p, latent: torch.Tensor, mean, std) -> torch.Tensor:
    """wan_wrapper.py:90-113: latent [1, F, 16, h, w] -> [1, T, 3, H, W] float32 in [-1, 1]."""
    zs = latent.permute(0, 2, 1, 3, 4)
    m = torch.tensor(mean, dtype=latent.dtype, device=latent.device)
    inv = 1.0 / torch.tensor(std, dtype=latent.dtype, device=latent.device)
    out = decode(p, zs, m, inv).float().clamp_(-1, 1)
    return out.permute(0, 2, 1, 3, 4)


def encode_to_latent(p, pixel: torch.Tensor, mean, std) -> torch.Tensor:
    """wan_wrapper.py:74-88: pixel [1, 3, T, H, W] -> [1, F, 16, h, w] float32."""
    m = torch.tensor(mean, dtype=pixel.dtype, device=pixel.device)
    inv = 1.0 / torch.tensor(std, dtype=pixel.dtype, device=pixel.device)
    return encode(p, pixel, m, inv).float().permute(0, 2, 1, 3, 4)
