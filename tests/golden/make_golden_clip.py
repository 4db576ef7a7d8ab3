"""CLIP vision tower fixture from the REAL reference (build container only): VisionTransformer.forward(x, use_31_block=True)
(MMPL_t2v/wan/modules/clip.py:209-327) in bf16 on seeded weights / pixels, reduced dims (head_dim stays 80 like ViT-H/14).
Writes tests/golden/clip_visual_tiny.pt."""
import importlib.util
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_import import REF, load_reference  # noqa: E402
from mmpl_amd.synthetic import clip_visual_state_dict, philox_normal  # noqa: E402
from oracle import clip_ref  # noqa: E402

META = dict(image_size=56, patch_size=14, dim=320, num_heads=4, num_layers=3, weight_seed=9, pixel_seed=51)


def load_clip_module():
    """clip.py imports torchvision.transforms, its tokenizer and XLM-R at module level; none of them is used by the vision tower"""
    _, _, attention, *_ = load_reference()
    for name in ("torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    for name, attrs in (("wan.modules.tokenizers", ["HuggingfaceTokenizer"]), ("wan.modules.xlm_roberta", ["XLMRoberta"])):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for a in attrs:
                setattr(m, a, type(a, (torch.nn.Module,), {}))
            sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("wan.modules.clip", os.path.join(REF, "wan", "modules", "clip.py"))
    clip = importlib.util.module_from_spec(spec)
    sys.modules["wan.modules.clip"] = clip
    spec.loader.exec_module(clip)
    # the reference's flash_attention keeps half-precision inputs in THEIR dtype (attention.py:68-75: only non-half inputs are
    # cast); its SDPA fallback would force bf16, so the dtype is passed through explicitly
    clip.flash_attention = lambda q, k, v, dropout_p=0.0, causal=False, version=None: attention.attention(q, k, v, dropout_p=dropout_p, causal=causal,
                                                                                                          dtype=q.dtype)
    return clip


def main():
    torch.set_grad_enabled(False)
    clip = load_clip_module()
    m = META
    vit = clip.VisionTransformer(image_size=m["image_size"], patch_size=m["patch_size"], dim=m["dim"], mlp_ratio=4, out_dim=64,
                                 num_heads=m["num_heads"], num_layers=m["num_layers"], pool_type="token", pre_norm=True, post_norm=False,
                                 activation="gelu", norm_eps=1e-5).eval()
    sd = clip_visual_state_dict(m["dim"], m["num_heads"], m["num_layers"], m["image_size"], m["patch_size"], seed=m["weight_seed"])
    ref = {k: tuple(v.shape) for k, v in vit.state_dict().items() if k != "head"}
    assert ref == {k: tuple(v.shape) for k, v in sd.items()}, set(ref) ^ set(sd)
    vit.load_state_dict(sd, strict=False)
    vit = vit.to(torch.bfloat16)
    # the reference's LayerNorm runs on x.float() (clip.py:45-48) with the weights cast up by autocast on its CUDA platform; the CPU
    # kernel refuses the mixed dtypes, so the (bf16-valued) norm parameters are held in fp32 here: same numbers, same arithmetic
    for mod in vit.modules():
        if isinstance(mod, torch.nn.LayerNorm):
            mod.float()
    x = philox_normal([2, 3, m["image_size"], m["image_size"]], m["pixel_seed"])
    out = vit(x, use_31_block=True)
    o = clip_ref.clip_visual(sd, x, m["num_heads"], m["num_layers"], m["patch_size"])
    print("[clip] out", tuple(out.shape), "rms", out.float().pow(2).mean().sqrt().item(), "oracle-vs-ref max|d|", (o.float() - out.float()).abs().max().item())
    # The dtype Wan-I2V really runs the tower in: fp16 weights under fp16 autocast (clip.py:509-511,540; wan_i2v_14B.py:17
    # clip_dtype = torch.float16), LayerNorm in fp32.  Same bf16-valued weights cast to fp16 (exact: bf16 -> fp16 only loses range),
    # same pixels: the distance of the bf16 product path to THIS output is the deviation the fixture above cannot show.
    vit16 = clip.VisionTransformer(image_size=m["image_size"], patch_size=m["patch_size"], dim=m["dim"], mlp_ratio=4, out_dim=64,
                                   num_heads=m["num_heads"], num_layers=m["num_layers"], pool_type="token", pre_norm=True, post_norm=False,
                                   activation="gelu", norm_eps=1e-5).eval()
    vit16.load_state_dict(sd, strict=False)
    vit16 = vit16.to(torch.float16)
    for mod in vit16.modules():
        if isinstance(mod, torch.nn.LayerNorm):
            mod.float()
    out16 = vit16(x.to(torch.float16), use_31_block=True)
    d = (out16.float() - out.float()).norm() / out16.float().norm()
    print("[clip] fp16 reference: rel_l2(bf16 reference, fp16 reference) =", d.item())
    torch.save(dict(out=out.clone(), out_fp16=out16.clone(), bf16_vs_fp16=float(d), meta=m), os.path.join(HERE, "clip_visual_tiny.pt"))


if __name__ == "__main__":
    main()
