#!/usr/bin/env python3
"""dev: sha256 of the large-GEMM outputs per (shape, epilogue, entry point).  Run once with MMPL_GEMM_V8=0 and once with =1 and diff the
two listings: gemm_bf16_v8_kernel issues the same MFMAs in the same per-accumulator order as v6, so every line must agree."""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmpl_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
BF = torch.bfloat16
shapes = [(25200, 15360, 5120, 0), (25200, 5120, 5120, 3), (21600, 13824, 5120, 1), (25200, 5120, 13824, 3), (7200, 5120, 5120, 1),
          (10920, 1536, 1536, 3), (10920, 1536, 8960, 3), (3120, 4608, 1536, 0), (9360, 1536, 1536, 4), (3120, 1536, 1536, 3), (10920, 4608, 1536, 0), (9360, 8960, 1536, 1), (1560, 1536, 1536, 2), (4096, 5120, 5120, 4), (1030, 264, 192, 2), (2000, 520, 128, 0)]
for M, N, K, epi in shapes:
    torch.manual_seed(M + N + K + epi)
    A = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    b = (torch.randn(N, device=dev) * 0.1).to(BF)
    res = torch.randn(M, N, device=dev).to(BF)
    gate = torch.randn((M + 3599) // 3600, N, device=dev).to(BF)
    ctr = torch.zeros(8, dtype=torch.int32, device=dev)
    nb = lib.mmpl_gemm_scratch_bytes()
    scratch = torch.zeros(nb, dtype=torch.uint8, device=dev)
    outs = []
    for which in ("plain", "tickets", "scratch"):
        out = torch.full((M, N), float("nan"), device=dev, dtype=BF)
        args = (_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(out), N, M, N, K, epi, _lib.ptr(res), N, _lib.ptr(gate), N, 3600)
        if which == "plain":
            _lib.check(lib.mmpl_gemm(*args, _lib.stream_ptr()))
        elif which == "tickets":
            _lib.check(lib.mmpl_gemm_tickets(*args, _lib.ptr(ctr), _lib.stream_ptr()))
        else:
            _lib.check(lib.mmpl_gemm_scratch(*args, _lib.ptr(scratch), nb, _lib.stream_ptr()))
        torch.cuda.synchronize()
        h = hashlib.sha256(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
        nan = int(torch.isnan(out.float()).sum())
        outs.append(f"{which} {h} nan={nan}")
    y = (A.float() @ W.float().t() + b.float()).to(BF)
    print(f"M={M} N={N} K={K} epi={epi}: " + " | ".join(outs) + f" | ctr {int(ctr.abs().sum())}", flush=True)
