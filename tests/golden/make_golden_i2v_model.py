"""Wan-I2V MODEL TYPE fixture from the REAL reference (build container only): WanModel(model_type='i2v')
(MMPL_t2v/wan/modules/model.py:500-760: in_dim 36, img_emb, WAN_CROSSATTENTION_CLASSES['i2v_cross_attn'] in every block)
run in bf16 on seeded weights / inputs, one forward over 3 latent frames.  Writes tests/golden/dit_i2v_tiny.pt."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import WAN_CONFIGS, dit_i2v_state_dict, philox_normal  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402

META = dict(cfg="tiny", weight_seed=5, F=3, lat_h=16, lat_w=16, t=673.0, n_valid_txt=40,
            seeds=dict(x=41, y=42, clip_fea=43, txt=44))


def inputs(meta=META):
    """(x [16,F,h,w], y [20,F,h,w], clip_fea [257,1280], context [n_valid, text_dim]) -- regenerated bit-identically by the tests"""
    cfg = WAN_CONFIGS[meta["cfg"]]
    F_, h, w, sd = meta["F"], meta["lat_h"], meta["lat_w"], meta["seeds"]
    x = philox_normal([16, F_, h, w], sd["x"])
    y = philox_normal([20, F_, h, w], sd["y"])
    clip_fea = philox_normal([257, 1280], sd["clip_fea"])
    txt = philox_normal([meta["n_valid_txt"], cfg["text_dim"]], sd["txt"])
    return x, y, clip_fea, txt


def main():
    torch.set_grad_enabled(False)
    _, model, attention, *_ = load_reference()
    model.flash_attention = attention.attention                  # the reference's own SDPA path (attention.py:170-185)
    cfg = WAN_CONFIGS[META["cfg"]]
    m = model.WanModel(model_type="i2v", in_dim=36, dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                       num_layers=cfg["num_layers"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"]).eval()
    sd = dit_i2v_state_dict(cfg, seed=META["weight_seed"])
    ref_keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert ref_keys == {k: tuple(v.shape) for k, v in sd.items()}, set(ref_keys) ^ set(sd)
    m.load_state_dict(sd, strict=True)
    m = m.to(torch.bfloat16)
    x, y, clip_fea, txt = inputs()
    F_, h, w = META["F"], META["lat_h"], META["lat_w"]
    S = (h // 2) * (w // 2)
    t = torch.tensor([META["t"]], dtype=torch.float32)
    out = m([x], t=t, context=[txt], seq_len=F_ * S, clip_fea=clip_fea.unsqueeze(0), y=[y])
    out = out[0] if isinstance(out, (list, tuple)) else out
    out = out.squeeze(0) if out.dim() == 5 else out                     # [16, F, h, w]
    # the oracle restatement of the same forward (full attention over the 3 frames = one stage that writes and sees all of them)
    ocfg = W.DitCfg(**dict(cfg, in_dim=36))
    kv = W.new_kv_cache(ocfg, F_, S)
    o = W.dit_forward(sd, ocfg, torch.cat([x, y], dim=0), torch.full([1, F_], META["t"], dtype=torch.float32), txt, kv, [None] * ocfg.num_layers,
                      list(range(F_)), list(range(F_)), list(range(F_)), clip_fea=clip_fea)
    print("[i2v model] out", tuple(out.shape), "rms", out.float().pow(2).mean().sqrt().item(), "oracle-vs-ref max|d|",
          (o.float() - out.float()).abs().max().item(), "rel_l2", ((o.float() - out.float()).norm() / out.float().norm()).item())
    torch.save(dict(out=out.clone(), meta=META), os.path.join(HERE, "dit_i2v_tiny.pt"))


if __name__ == "__main__":
    main()
