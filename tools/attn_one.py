"""One attention launch population for rocprofv3 --pmc (dev tool): python3 tools/attn_one.py [stage] [iters]
Runs the op the DiT forward runs: the w64 kernel on a producer-prescaled q, with the split-KV workspace (variant 4)."""
import ctypes as C
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmpl_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
BF = torch.bfloat16
stage = sys.argv[1] if len(sys.argv) > 1 else "s1"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq, npg = {"s0": (2, 2), "s1": (7, 9), "s2": (6, 13), "s3": (6, 21)}[stage]
H, S, d = 40, 3600, 5120
kc = torch.randn(npg * S, d, device=dev).to(BF)
vc = torch.randn(npg * S, d, device=dev).to(BF)
Lq = nq * S
q = (torch.randn(Lq, 3 * d, device=dev) * (1.4426950408889634 / math.sqrt(128))).to(BF)
o = torch.empty(Lq, d, device=dev, dtype=BF)
ws = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev)
kp = (C.c_void_p * npg)(*[kc[i * S:].data_ptr() for i in range(npg)])
vp = (C.c_void_p * npg)(*[vc[i * S:].data_ptr() for i in range(npg)])

for _ in range(iters):
    _lib.check(lib.mmpl_attn_fwd_variant(_lib.ptr(q), 3 * d, _lib.ptr(o), d, kp, vp, d, d, npg, S, Lq, H, 1 / math.sqrt(128),
                                         _lib.ptr(ws), ws.numel(), 4, 0, _lib.stream_ptr()))
torch.cuda.synchronize()
print("done", stage, float(o.float().abs().mean()))
