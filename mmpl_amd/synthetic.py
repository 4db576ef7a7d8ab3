"""Seeded synthetic Wan2.1 DiT / VAE weights in the reference's state_dict key layout.

There are no checkpoints in the build container or on the GPU box, so parity tests and ``bench.py`` run on
weights drawn here.  Shapes and key names are exactly those of ``CausalFPSWanModel.state_dict()``
(MMPL_t2v/wan/modules/causal_fps_model.py:485-506) so a real Wan2.1 / MMPL checkpoint binds the same way.
The distributions follow ``init_weights`` (causal_fps_model.py:1032-1054: Xavier-uniform linears,
N(0, 0.02) embeddings) except that biases, norm gains and the (zero-initialised) head are made non-trivial
so that every term of the forward is exercised.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict

import torch

WAN_CONFIGS: Dict[str, dict] = {
    # MMPL_t2v/wan/configs/wan_t2v_14B.py:17-25, wan_t2v_1_3B.py:17-25
    "14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, text_dim=4096, freq_dim=256),
    "1.3B": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, text_dim=4096, freq_dim=256),
    # reduced configs for parity tests (head_dim stays 128 like both real models)
    "tiny": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, freq_dim=256),
    "small": dict(dim=512, ffn_dim=1280, num_heads=4, num_layers=3, text_dim=128, freq_dim=256),
    # the depth at which the 408-forward trajectory tolerance is stated a second time (tests/golden/make_golden.py chunk50_deep)
    "deep": dict(dim=512, ffn_dim=1536, num_heads=4, num_layers=8, text_dim=128, freq_dim=256),
}


def dit_state_dict(cfg: dict, seed: int = 0, dtype=torch.bfloat16, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    dim, ffn, L = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    text_dim, freq_dim = cfg.get("text_dim", 4096), cfg.get("freq_dim", 256)
    in_dim, out_dim = cfg.get("in_dim", 16), cfg.get("out_dim", 16)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def randn(*shape, std=1.0):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def xavier(out_f, in_f):
        a = math.sqrt(6.0 / (in_f + out_f))
        return ((torch.rand(out_f, in_f, generator=g, device=device, dtype=torch.float32) * 2 - 1) * a).to(dtype)

    def linear(name, out_f, in_f, std=None):
        sd[name + ".weight"] = xavier(out_f, in_f) if std is None else randn(out_f, in_f, std=std)
        sd[name + ".bias"] = randn(out_f, std=0.02)

    sd["patch_embedding.weight"] = xavier(dim, in_dim * 4).view(dim, in_dim, 1, 2, 2).contiguous()
    sd["patch_embedding.bias"] = randn(dim, std=0.02)
    linear("text_embedding.0", dim, text_dim, std=0.02)
    linear("text_embedding.2", dim, dim, std=0.02)
    linear("time_embedding.0", dim, freq_dim, std=0.02)
    linear("time_embedding.2", dim, dim, std=0.02)
    linear("time_projection.1", 6 * dim, dim)
    for i in range(L):
        p = f"blocks.{i}."
        sd[p + "modulation"] = randn(1, 6, dim, std=dim ** -0.5)
        for a in ("self_attn", "cross_attn"):
            for n in ("q", "k", "v", "o"):
                linear(p + f"{a}.{n}", dim, dim)
            sd[p + f"{a}.norm_q.weight"] = (1 + randn(dim, std=0.1).float()).to(dtype)
            sd[p + f"{a}.norm_k.weight"] = (1 + randn(dim, std=0.1).float()).to(dtype)
        sd[p + "norm3.weight"] = (1 + randn(dim, std=0.1).float()).to(dtype)
        sd[p + "norm3.bias"] = randn(dim, std=0.02)
        linear(p + "ffn.0", ffn, dim)
        linear(p + "ffn.2", dim, ffn)
    sd["head.modulation"] = randn(1, 2, dim, std=dim ** -0.5)
    sd["head.head.weight"] = randn(out_dim * 4, dim, std=0.02)
    sd["head.head.bias"] = randn(out_dim * 4, std=0.02)
    return sd


def dit_i2v_state_dict(cfg: dict, seed: int = 0, clip_dim: int = 1280, dtype=torch.bfloat16, device="cpu"):
    """state_dict of WanModel(model_type='i2v') (MMPL_t2v/wan/modules/model.py:563-616): the t2v keys with in_dim 36, plus
    ``img_emb`` (MLPProj) and every block's ``cross_attn.{k_img, v_img, norm_k_img}``."""
    cfg = dict(cfg, in_dim=cfg.get("in_dim", 36))
    sd = dit_state_dict({k: v for k, v in cfg.items() if k != "model_type"}, seed=seed, dtype=dtype, device=device)
    dim = cfg["dim"]
    for i in range(cfg["num_layers"]):
        ca, _ = i2v_cross_state_dict(dim, clip_dim, seed=seed * 1000 + 17 + i, dtype=dtype, device=device)
        for k in ("k_img.weight", "k_img.bias", "v_img.weight", "v_img.bias", "norm_k_img.weight"):
            sd[f"blocks.{i}.cross_attn.{k}"] = ca[k]
    _, mp = i2v_cross_state_dict(dim, clip_dim, seed=seed * 1000 + 7, dtype=dtype, device=device)
    for k, v in mp.items():
        sd["img_emb." + k] = v
    return sd


def clip_visual_state_dict(dim: int, num_heads: int, num_layers: int, image_size: int = 224, patch_size: int = 14, mlp_ratio: int = 4,
                           seed: int = 0, dtype=torch.bfloat16, device="cpu"):
    """state_dict of the reference's VisionTransformer(pool_type='token', pre_norm=True) minus ``head`` (a plain Parameter the
    use_31_block path never touches): keys / shapes of MMPL_t2v/wan/modules/clip.py:253-286."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

    def norm(name):
        sd[name + ".weight"] = (1 + rn(dim, std=0.1).float()).to(dtype)
        sd[name + ".bias"] = rn(dim, std=0.02)

    def lin(name, out_f, in_f):
        sd[name + ".weight"] = rn(out_f, in_f, std=in_f ** -0.5)
        sd[name + ".bias"] = rn(out_f, std=0.02)
    sd = OrderedDict()
    n_patch = (image_size // patch_size) ** 2
    sd["cls_embedding"] = rn(1, 1, dim, std=dim ** -0.5)
    sd["pos_embedding"] = rn(1, n_patch + 1, dim, std=dim ** -0.5)
    sd["patch_embedding.weight"] = rn(dim, 3, patch_size, patch_size, std=(3 * patch_size * patch_size) ** -0.5)
    norm("pre_norm")
    for i in range(num_layers):
        p = f"transformer.{i}."
        norm(p + "norm1")
        lin(p + "attn.to_qkv", 3 * dim, dim)
        lin(p + "attn.proj", dim, dim)
        norm(p + "norm2")
        lin(p + "mlp.0", int(dim * mlp_ratio), dim)
        lin(p + "mlp.2", dim, int(dim * mlp_ratio))
    norm("post_norm")
    return sd


def philox_normal(shape, seed: int, dtype=torch.bfloat16) -> torch.Tensor:
    """Build-owned counter-based N(0,1) generator (numpy Philox) so CPU container and GPU box regenerate
    identical inputs without the reference (RNG note)."""
    import numpy as np
    rng = np.random.Generator(np.random.Philox(key=seed))
    n = 1
    for s in shape:
        n *= int(s)
    return torch.from_numpy(rng.standard_normal(n, dtype=np.float32)).reshape(*shape).to(dtype)


# ---------------------------------------------------------------------------------------------- Wan 3D-VAE weights
VAE_DIM, VAE_Z, VAE_DIM_MULT, VAE_T_DOWN = 96, 16, [1, 2, 4, 4], [False, True, True]


def vae_layout():
    """(key, shape) of every tensor of `WanVAE_(dim=96, z_dim=16, ...)` (MMPL_t2v/wan/modules/vae.py:612-636), in
    state_dict order.  Derived from the constructor logic (vae.py:284-316, 388-421), not from a checkpoint."""
    out = []

    def conv3(name, cout, cin, k):
        out.append((name + ".weight", (cout, cin) + tuple(k)))
        out.append((name + ".bias", (cout,)))

    def res(pre, cin, cout):
        out.append((pre + "residual.0.gamma", (cin, 1, 1, 1)))
        conv3(pre + "residual.2", cout, cin, (3, 3, 3))
        out.append((pre + "residual.3.gamma", (cout, 1, 1, 1)))
        conv3(pre + "residual.6", cout, cout, (3, 3, 3))
        if cin != cout:
            conv3(pre + "shortcut", cout, cin, (1, 1, 1))

    def attn(pre, c):
        out.append((pre + "norm.gamma", (c, 1, 1)))
        conv3(pre + "to_qkv", 3 * c, c, (1, 1))
        conv3(pre + "proj", c, c, (1, 1))

    # encoder
    dims = [VAE_DIM * u for u in [1] + VAE_DIM_MULT]
    conv3("encoder.conv1", dims[0], 3, (3, 3, 3))
    j = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(2):
            res(f"encoder.downsamples.{j}.", cin, cout)
            cin = cout
            j += 1
        if i != len(VAE_DIM_MULT) - 1:
            conv3(f"encoder.downsamples.{j}.resample.1", cout, cout, (3, 3))
            if VAE_T_DOWN[i]:
                conv3(f"encoder.downsamples.{j}.time_conv", cout, cout, (3, 1, 1))
            j += 1
    res("encoder.middle.0.", dims[-1], dims[-1])
    attn("encoder.middle.1.", dims[-1])
    res("encoder.middle.2.", dims[-1], dims[-1])
    out.append(("encoder.head.0.gamma", (dims[-1], 1, 1, 1)))
    conv3("encoder.head.2", 2 * VAE_Z, dims[-1], (3, 3, 3))
    conv3("conv1", 2 * VAE_Z, 2 * VAE_Z, (1, 1, 1))
    conv3("conv2", VAE_Z, VAE_Z, (1, 1, 1))
    # decoder
    dims = [VAE_DIM * u for u in [VAE_DIM_MULT[-1]] + VAE_DIM_MULT[::-1]]
    t_up = VAE_T_DOWN[::-1]
    conv3("decoder.conv1", dims[0], VAE_Z, (3, 3, 3))
    res("decoder.middle.0.", dims[0], dims[0])
    attn("decoder.middle.1.", dims[0])
    res("decoder.middle.2.", dims[0], dims[0])
    j = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(3):
            res(f"decoder.upsamples.{j}.", cin, cout)
            cin = cout
            j += 1
        if i != len(VAE_DIM_MULT) - 1:
            conv3(f"decoder.upsamples.{j}.resample.1", cout // 2, cout, (3, 3))
            if t_up[i]:
                conv3(f"decoder.upsamples.{j}.time_conv", cout * 2, cout, (3, 1, 1))
            j += 1
    out.append(("decoder.head.0.gamma", (dims[-1], 1, 1, 1)))
    conv3("decoder.head.2", 3, dims[-1], (3, 3, 3))
    return out


def vae_state_dict(seed: int = 0, dtype=torch.bfloat16, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in vae_layout():
        if name.endswith(".gamma"):
            t = 1 + 0.1 * torch.randn(*shape, generator=g, device=device)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(*shape, generator=g, device=device)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(*shape, generator=g, device=device) * (1.0 / math.sqrt(fan_in))
        sd[name] = t.to(dtype)
    return sd


# ---------------------------------------------------------------------------------------------- umT5 encoder weights
T5_CONFIGS = {
    "umt5-xxl": dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32),
    "tiny": dict(vocab=1000, dim=256, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=2, num_buckets=32),
    "small": dict(vocab=4000, dim=512, dim_attn=512, dim_ffn=1280, num_heads=8, num_layers=3, num_buckets=32),
}


def t5_state_dict(cfg: dict, seed: int = 0, dtype=torch.bfloat16, device="cpu") -> "OrderedDict[str, torch.Tensor]":
    """Keys / shapes of `T5Encoder.state_dict()` (MMPL_t2v/wan/modules/t5.py:267-312, shared_pos=False); init as t5.py:27-42."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d, da, df, n, nb = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_buckets"]

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g, device=device) * std).to(dtype)

    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    sd["token_embedding.weight"] = rn(cfg["vocab"], d, std=1.0)
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = (1 + 0.1 * torch.randn(d, generator=g, device=device)).to(dtype)
        sd[p + "attn.q.weight"] = rn(da, d, std=(d * (da // n)) ** -0.5 * 4)
        sd[p + "attn.k.weight"] = rn(da, d, std=d ** -0.5)
        sd[p + "attn.v.weight"] = rn(da, d, std=d ** -0.5)
        sd[p + "attn.o.weight"] = rn(d, da, std=da ** -0.5)
        sd[p + "norm2.weight"] = (1 + 0.1 * torch.randn(d, generator=g, device=device)).to(dtype)
        sd[p + "ffn.gate.0.weight"] = rn(df, d, std=d ** -0.5)
        sd[p + "ffn.fc1.weight"] = rn(df, d, std=d ** -0.5)
        sd[p + "ffn.fc2.weight"] = rn(d, df, std=df ** -0.5)
        sd[p + "pos_embedding.embedding.weight"] = rn(nb, n, std=0.5)
    sd["norm.weight"] = (1 + 0.1 * torch.randn(d, generator=g, device=device)).to(dtype)
    return sd


def i2v_cross_state_dict(dim: int, clip_dim: int = 1280, seed: int = 0, dtype=torch.bfloat16, device="cpu"):
    """(WanI2VCrossAttention state_dict, MLPProj state_dict): keys / shapes of MMPL_t2v/wan/modules/model.py:224-236, 469-477."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def rn(*shape, std):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(dtype)
    ca = OrderedDict()
    for m in ("q", "k", "v", "o", "k_img", "v_img"):
        ca[m + ".weight"] = rn(dim, dim, std=dim ** -0.5)
        ca[m + ".bias"] = rn(dim, std=0.02)
    for m in ("norm_q", "norm_k", "norm_k_img"):
        ca[m + ".weight"] = (1 + rn(dim, std=0.1).float()).to(dtype)
    mp = OrderedDict()
    mp["proj.0.weight"] = (1 + rn(clip_dim, std=0.1).float()).to(dtype)
    mp["proj.0.bias"] = rn(clip_dim, std=0.02)
    mp["proj.1.weight"] = rn(clip_dim, clip_dim, std=clip_dim ** -0.5)
    mp["proj.1.bias"] = rn(clip_dim, std=0.02)
    mp["proj.3.weight"] = rn(dim, clip_dim, std=clip_dim ** -0.5)
    mp["proj.3.bias"] = rn(dim, std=0.02)
    mp["proj.4.weight"] = (1 + rn(dim, std=0.1).float()).to(dtype)
    mp["proj.4.bias"] = rn(dim, std=0.02)
    return ca, mp
