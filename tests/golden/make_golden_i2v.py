"""Wan-I2V image cross-attention fixture from the REAL reference modules (build container only): MLPProj and
WanI2VCrossAttention run standalone in bf16 on seeded weights / inputs."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import i2v_cross_state_dict, philox_normal  # noqa: E402
from oracle import i2v_ref  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    _, model, attention, *_ = load_reference()
    model.flash_attention = attention.attention                  # WanI2VCrossAttention -> the reference's own SDPA path
    dim, heads, Lq = 256, 2, 200
    ca_sd, mp_sd = i2v_cross_state_dict(dim, seed=6)
    ca = model.WanI2VCrossAttention(dim, heads).eval()
    assert sorted((k, tuple(v.shape)) for k, v in ca.state_dict().items()) == sorted((k, tuple(v.shape)) for k, v in ca_sd.items())
    ca.load_state_dict(ca_sd)
    ca = ca.to(torch.bfloat16)
    mp = model.MLPProj(1280, dim).eval()
    assert sorted((k, tuple(v.shape)) for k, v in mp.state_dict().items()) == sorted((k, tuple(v.shape)) for k, v in mp_sd.items())
    mp.load_state_dict(mp_sd)
    mp = mp.to(torch.bfloat16)
    clip_fea = philox_normal([1, 257, 1280], 31)
    txt = philox_normal([1, 512, dim], 32)
    txt[:, 40:] = 0
    x = philox_normal([1, Lq, dim], 33)
    ctx_img = mp(clip_fea)
    context = torch.cat([ctx_img, txt], dim=1)
    out = ca(x, context, None)
    o_img = i2v_ref.mlp_proj(mp_sd, clip_fea)
    o_out = i2v_ref.i2v_cross_attention(ca_sd, x, torch.cat([o_img, txt], dim=1), heads)
    print("[i2v] oracle-vs-ref max|d|: proj", (o_img.float() - ctx_img.float()).abs().max().item(), "cross-attn",
          (o_out.float() - out.float()).abs().max().item(), "rms", out.float().pow(2).mean().sqrt().item())
    # inputs are philox-seeded (regenerated bit-identically by the tests): only the reference's outputs are stored
    torch.save(dict(ctx_img=ctx_img.clone(), out=out.clone(),
                    meta=dict(dim=dim, heads=heads, Lq=Lq, weight_seed=6, seeds=dict(clip_fea=31, txt=32, x=33), n_valid_txt=40)), os.path.join(HERE, "i2v_cross_tiny.pt"))


if __name__ == "__main__":
    main()
