"""umT5 encoder fixture from the REAL reference (build container only): tiny config, bf16, seeded weights / ids."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from _ref_import import load_reference  # noqa: E402
from mmpl_amd.synthetic import T5_CONFIGS, t5_state_dict  # noqa: E402
from oracle import t5_ref  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    load_reference()
    # wan.modules.t5 imports .tokenizers (ftfy, absent): stub it -- only the encoder module is exercised
    sys.modules["wan.modules.tokenizers"] = types.SimpleNamespace(HuggingfaceTokenizer=None)
    import importlib
    _cd = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0          # t5.py:481 evaluates it as a default argument at import time
    try:
        t5 = importlib.import_module("wan.modules.t5")
    finally:
        torch.cuda.current_device = _cd
    cfg = T5_CONFIGS["tiny"]
    m = t5.T5Encoder(cfg["vocab"], cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_layers"], cfg["num_buckets"],
                     shared_pos=False, dropout=0.1).eval()
    sd = t5_state_dict(cfg, seed=4)
    ref_keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
    assert sorted(ref_keys) == sorted((k, tuple(v.shape)) for k, v in sd.items()), "t5_state_dict must match the reference layout"
    m.load_state_dict(sd)
    m32 = m.float()                         # the wrapper's own mode: bf16 checkpoint upcast to fp32 on the CPU (wan_wrapper.py:19-27)
    L = 128
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(2, cfg["vocab"], (2, L), generator=g)
    mask = torch.zeros(2, L, dtype=torch.long)
    for b, n in enumerate((37, 128)):
        mask[b, :n] = 1
        ids[b, n:] = 0
    out_f32 = m32(ids, mask)
    m = m32.to(torch.bfloat16)              # what the HIP engine computes in (and the reference's generate.py path, t5.py:470-513)
    out = m(ids, mask)
    for o_ in (out, out_f32):
        for u, v in zip(o_, mask.gt(0).sum(dim=1).long()):
            u[v:] = 0.0                                                    # wan_wrapper.py:46-47
    sd32 = {k: v.float() for k, v in sd.items()}
    o32 = t5_ref.text_encoder_forward(sd32, ids, mask, cfg["num_heads"], cfg["num_buckets"], cfg["num_layers"])
    print("[t5] fp32 oracle-vs-ref max|d| =", (o32 - out_f32).abs().max().item(), " bf16-vs-fp32 ref rel_l2 =",
          ((out.float() - out_f32).norm() / out_f32.norm()).item())
    o = t5_ref.text_encoder_forward(sd, ids, mask, cfg["num_heads"], cfg["num_buckets"], cfg["num_layers"])
    print("[t5] oracle-vs-ref max|d| =", (o.float() - out.float()).abs().max().item(), "rms", out.float().pow(2).mean().sqrt().item())
    torch.save(dict(ids=ids, mask=mask, out=out.clone(), out_f32=out_f32.clone(), meta=dict(cfg="tiny", weight_seed=4, L=L)), os.path.join(HERE, "t5_tiny.pt"))


if __name__ == "__main__":
    main()
