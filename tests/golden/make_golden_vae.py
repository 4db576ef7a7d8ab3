"""VAE fixtures from the REAL reference (build container only): tiny-spatial decode / encode in bf16, the
prefix-equivalence facts the hand-off relies on, and the real left-over hand-off tensor as a realistic input."""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import philox_normal, vae_layout, vae_state_dict  # noqa: E402
from oracle import vae_ref  # noqa: E402

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
        -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253,
       2.8251, 1.9160]


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def gen_vae():
    torch.set_grad_enabled(False)
    *_, vae, _, _ = load_reference()[:4], None, None
    vae = load_reference()[3]
    m = vae.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[], temperal_downsample=[False, True, True],
                    dropout=0.0).eval()
    ref_keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert ref_keys == vae_layout(), "synthetic.vae_layout() must reproduce the reference state_dict layout"
    sd = vae_state_dict(seed=3)
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.bfloat16)
    mean, inv = torch.tensor(MEAN, dtype=torch.bfloat16), 1.0 / torch.tensor(STD, dtype=torch.bfloat16)
    out = {}
    # ---- decode: 4 latent frames of 8x12 -> 13 pixel frames of 64x96
    z = philox_normal([1, 16, 4, 8, 12], 31)
    t0 = time.time()
    px = m.decode(z, [mean, inv])
    po = vae_ref.decode(sd, z, mean, inv)
    print(f"[vae] decode ref {time.time() - t0:.1f}s  oracle-vs-ref rel_l2 = {rel_l2(po, px):.3e}  out rms={px.float().pow(2).mean().sqrt():.3f}")
    out["dec_out"] = px.clone()
    # prefix equivalence (causal VAE): decode of the first 2 latents == first 5 frames of the 4-latent decode
    px2 = m.decode(z[:, :, :2], [mean, inv])
    out["dec_prefix_equal"] = bool(torch.equal(px2, px[:, :, :5]))
    print("[vae] decode prefix-equivalence (bf16):", out["dec_prefix_equal"])
    # ---- encode: 9 pixel frames of 64x96 -> 3 latents
    x = philox_normal([1, 3, 9, 64, 96], 32).clamp(-1, 1)
    t0 = time.time()
    lat = m.encode(x, [mean, inv])
    lo = vae_ref.encode(sd, x, mean, inv)
    print(f"[vae] encode ref {time.time() - t0:.1f}s  oracle-vs-ref rel_l2 = {rel_l2(lo, lat):.3e}")
    out["enc_out"] = lat.clone()
    lat2 = m.encode(x[:, :, :5], [mean, inv])
    out["enc_prefix_equal"] = bool(torch.equal(lat2, lat[:, :, :2]))
    print("[vae] encode prefix-equivalence (bf16):", out["enc_prefix_equal"])
    out["meta"] = dict(weight_seed=3, z_seed=31, x_seed=32, z_shape=[1, 16, 4, 8, 12], x_shape=[1, 3, 9, 64, 96])
    torch.save(out, os.path.join(HERE, "vae_tiny.pt"))
    # the real hand-off artefact (bf16 [1,3,16,60,104]) is DATA the reference tree holds; keep a spatial crop as a
    # realistic-statistics input fixture
    art = torch.load("/root/reference/MMPL_i2v/latents_chunk4.pt", map_location="cpu")
    print("[vae] artefact", art.shape, art.dtype, float(art.float().mean()), float(art.float().std()))
    torch.save(art[..., :16, :24].contiguous().clone(), os.path.join(HERE, "handoff_artefact_crop.pt"))


if __name__ == "__main__":
    gen_vae()
