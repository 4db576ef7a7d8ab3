#!/usr/bin/env python3
"""Dev tool: which kv rows does an attention variant actually see?  V = identity-coded rows (S <= 128 per page, one page), so
the output row IS the softmax weight vector; random q, k.    python tools/attn_rows.py Lq H S [variants...]"""
import ctypes as C
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.attn_dev import run, dev, BF  # noqa: E402

Lq, H, S = (int(x) for x in sys.argv[1:4])
variants = [int(x) for x in sys.argv[4:]] or [1, 3]
assert S <= 128
torch.manual_seed(1)
d = H * 128
q = torch.randn(Lq, d, device=dev).to(BF)
k = torch.randn(S, d, device=dev).to(BF)
v = torch.zeros(S, d, device=dev)
for r in range(S):
    v[r, r::128] = 1.0
v = v.to(BF)
kp = (C.c_void_p * 1)(k.data_ptr()); vp = (C.c_void_p * 1)(v.data_ptr())
qq = q.float().view(Lq, H, 128).transpose(0, 1); kk = k.float().view(S, H, 128).transpose(0, 1)
ref = torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(128), -1)            # H, Lq, S
for var in variants:
    o = torch.zeros(Lq, d, device=dev, dtype=BF)
    run(var, q, d, o, d, kp, vp, 1, S, Lq, H)
    torch.cuda.synchronize()
    w = o.float().view(Lq, H, 128).transpose(0, 1)[:, :, :S]
    bad = ((w - ref).abs() > 0.03 * ref.amax(-1, keepdim=True) + 1e-3)
    print(f"v{var}: bad entries {int(bad.sum())} of {bad.numel()}; per head {bad.sum((1, 2)).tolist()}")
    if bad.any():
        rows = bad.any(-1).nonzero()
        print("   bad (head,q row) count", len(rows), "first", rows[:12].tolist())
        kv = bad.any(1).nonzero()
        print("   bad (head,kv row):", kv.tolist()[:80])
        h, r = rows[0].tolist()
        cols = bad[h, r].nonzero().flatten().tolist()
        print(f"   head {h} q {r}: got/ref", [(c, round(w[h, r, c].item(), 4), round(ref[h, r, c].item(), 4)) for c in cols[:24]])
