#!/usr/bin/env python3
"""sha256 of attn_w64_kernel's output (stateless launch, prescaled q) on fixed seeded inputs at a few shapes -- to tell whether two
builds of the library compute the same BITS (tools/r06_gpu.sh attnhash: two prebuilt libraries)."""
import ctypes as C
import hashlib
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (bound by hand, not through mmpl_amd._lib: an older library lacks symbols that module binds)
lib = C.CDLL(os.path.join(ROOT, "mmpl_amd", "lib", "libmmpl_hip.so"))
vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
lib.mmpl_attn_fwd_variant.argtypes = [vp, ci, vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ci, ci, ci, ci, ci, cf, vp, sz, ci, ci, vp]
lib.mmpl_attn_workspace_bytes.restype = sz
lib.mmpl_last_error.restype = C.c_char_p
dev = "cuda:0"
for name, (Lq, H, S, n_pages, gain) in {"small": (700, 2, 328, 3, 1.0), "one_block": (192, 2, 96, 2, 1.0), "s1_1p3B": (10920, 12, 1560, 9, 1.0),
                                        "heavy": (2600, 8, 640, 3, 6.0), "split_tail": (256 * 41 - 57, 8, 640, 3, 1.0)}.items():
    torch.manual_seed(7)
    d = H * 128
    c = (1.0 / math.sqrt(128)) * 1.4426950408889634
    q = (torch.randn(Lq, d, device=dev) * gain * c).to(torch.bfloat16)
    kc = (torch.randn(n_pages * S, d, device=dev) * gain).to(torch.bfloat16)
    vc = torch.randn(n_pages * S, d, device=dev).to(torch.bfloat16)
    o = torch.zeros(Lq, d, device=dev, dtype=torch.bfloat16)
    kp = (C.c_void_p * n_pages)(*[kc[i * S:].data_ptr() for i in range(n_pages)])
    vpp = (C.c_void_p * n_pages)(*[vc[i * S:].data_ptr() for i in range(n_pages)])
    ws = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev)
    rc = lib.mmpl_attn_fwd_variant(q.data_ptr(), d, o.data_ptr(), d, kp, vpp, d, d, n_pages, S, Lq, H, 1.0 / math.sqrt(128), ws.data_ptr(), ws.numel(), 4, 0,
                                   torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.mmpl_last_error()
    torch.cuda.synchronize()
    print("attnhash", name, hashlib.sha256(o.cpu().view(torch.uint8).numpy().tobytes()).hexdigest()[:16], flush=True)
