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


def philox_normal(shape, seed: int, dtype=torch.bfloat16) -> torch.Tensor:
    """Build-owned counter-based N(0,1) generator (numpy Philox) so CPU container and GPU box regenerate
    identical inputs without the reference (RNG note)."""
    import numpy as np
    rng = np.random.Generator(np.random.Philox(key=seed))
    n = 1
    for s in shape:
        n *= int(s)
    return torch.from_numpy(rng.standard_normal(n, dtype=np.float32)).reshape(*shape).to(dtype)
