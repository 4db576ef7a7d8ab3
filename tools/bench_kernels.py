#!/usr/bin/env python3
"""Micro-benchmarks of the two MFMA kernels on the 14B/720p shapes (dev tool).
    python tools/bench_kernels.py attn|gemm|all|gemmref|attnref|elem|cross [--iters N]"""
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


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench_attn(iters):
    H, S, d = 40, 3600, 5120
    n_slots = 21
    kc = torch.randn(n_slots * S, d, device=dev).to(BF)
    vc = torch.randn(n_slots * S, d, device=dev).to(BF)
    for name, nq, npg in (("s0", 2, 2), ("s1", 7, 9), ("s2", 6, 13), ("s3", 6, 21), ("cross", 7, 0)):
        Lq = nq * S
        q = torch.randn(Lq, 3 * d, device=dev).to(BF)
        o = torch.empty(Lq, d, device=dev, dtype=BF)
        if npg == 0:
            pr, npg_ = 512, 1
        else:
            pr, npg_ = S, npg
        kp = (C.c_void_p * npg_)(*[kc[i * S:].data_ptr() for i in range(npg_)])
        vp = (C.c_void_p * npg_)(*[vc[i * S:].data_ptr() for i in range(npg_)])
        fn = lambda: _lib.check(lib.mmpl_attn_fwd(_lib.ptr(q), 3 * d, _lib.ptr(o), d, kp, vp, d, d, npg_, pr, Lq, H,
                                                  1 / math.sqrt(128), _lib.stream_ptr()))
        ms = timeit(fn, iters)
        fl = 4.0 * Lq * npg_ * pr * d
        print(f"attn {name}: Lq={Lq} Lkv={npg_ * pr}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)


def bench_attn_ref(iters):
    """yardstick only (never on the product path): PyTorch-ROCm's scaled_dot_product_attention (its flash / memory-efficient backends:
    AOTriton / CK as the wheel was built) on the self-attention shapes of the 14B / 720p stages, K / V gathered CONTIGUOUS (what the
    reference materialises per layer before calling flash-attn, causal_fps_model.py:219-227) -- the gather itself is not timed.
    Next to it: mmpl_attn_fwd on the same q / k / v values (paged, in place)."""
    import torch.nn.functional as F
    from torch.nn.attention import SDPBackend, sdpa_kernel
    H, S, d = 40, 3600, 5120
    for name, nq, nkv in (("s0", 2, 2), ("s1", 7, 9), ("s2", 6, 13), ("s3", 6, 21)):
        Lq, Lkv = nq * S, nkv * S
        q = torch.randn(Lq, d, device=dev).to(BF)
        k = torch.randn(Lkv, d, device=dev).to(BF)
        v = torch.randn(Lkv, d, device=dev).to(BF)
        fl = 4.0 * Lq * Lkv * d
        o = torch.empty(Lq, d, device=dev, dtype=BF)
        kp = (C.c_void_p * nkv)(*[k[i * S:].data_ptr() for i in range(nkv)])
        vp = (C.c_void_p * nkv)(*[v[i * S:].data_ptr() for i in range(nkv)])
        ws = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev)
        fn = lambda: _lib.check(lib.mmpl_attn_fwd_variant(_lib.ptr(q), d, _lib.ptr(o), d, kp, vp, d, d, nkv, S, Lq, H, 1 / math.sqrt(128),
                                                          _lib.ptr(ws), ws.numel(), 3, 0, _lib.stream_ptr()))
        ms = min(timeit(fn, iters), timeit(fn, iters))
        print(f"attnref {name}: Lq={Lq} Lkv={Lkv}  mmpl attn_w64 (paged, raw q) {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
        # [B=1, H, L, 128] views of the token-major tensors (what flash-attn style kernels take) and head-major contiguous copies
        for layout in ("token-major views", "head-major contiguous"):
            q4, k4, v4 = (t.view(1, -1, H, 128).transpose(1, 2) for t in (q, k, v))
            if layout.startswith("head"):
                q4, k4, v4 = q4.contiguous(), k4.contiguous(), v4.contiguous()
            for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
                try:
                    with sdpa_kernel([be]):
                        f2 = lambda: F.scaled_dot_product_attention(q4, k4, v4)
                        ref = f2()
                        ms2 = min(timeit(f2, iters), timeit(f2, iters))
                    err = ((ref.transpose(1, 2).reshape(Lq, d).float() - o.float()).norm() / o.float().norm()).item()
                    print(f"attnref {name}: torch SDPA {be.name:22s} {layout:22s} {ms2:8.3f} ms  {fl / ms2 / 1e9:8.1f} TFLOP/s   (rel-L2 vs mmpl {err:.2e})", flush=True)
                    del ref
                except Exception as e:
                    print(f"attnref {name}: torch SDPA {be.name} {layout}: unavailable ({str(e).splitlines()[0][:120]})", flush=True)
            del q4, k4, v4
        del q, k, v, o


def bench_gemm(iters):
    m_big = int(os.environ.get("BENCH_M", "25200"))          # the stage's query rows: 25200 (s1), 21600 (s2, s3)
    shapes = (("qkv", m_big, 15360, 5120, 0), ("o", m_big, 5120, 5120, 3), ("ffn0", m_big, 13824, 5120, 1),
              ("ffn2", m_big, 5120, 13824, 3), ("qkv_s0", 7200, 15360, 5120, 0), ("sq8k", 8192, 8192, 8192, 0))
    if os.environ.get("BENCH_SHAPES"):                       # "name:M:N:K:epi,..." (e.g. the 1.3B / 480p block shapes)
        shapes = tuple((f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])) for f in (x.split(":") for x in os.environ["BENCH_SHAPES"].split(",")))
    for name, M, N, K, epi in shapes:
        # BENCH_PAD_K / BENCH_PAD_N: extra elements in the leading dimension of the K-contiguous operands / of C and the residual
        # (dev: does the power-of-two-ish row stride camp on a few HBM channels?)
        pk, pn = int(os.environ.get("BENCH_PAD_K", "0")), int(os.environ.get("BENCH_PAD_N", "0"))
        A = torch.randn(M, K + pk, device=dev).to(BF)
        W = (torch.randn(N, K + pk, device=dev) / math.sqrt(K)).to(BF)
        b = torch.randn(N, device=dev).to(BF)
        Cc = torch.empty(M, N + pn, device=dev, dtype=BF)
        res = torch.randn(M, N + pn, device=dev).to(BF)
        gate = torch.randn(8, N, device=dev).to(BF)
        fn = lambda: _lib.check(lib.mmpl_gemm(_lib.ptr(A), K + pk, _lib.ptr(W), K + pk, _lib.ptr(b), _lib.ptr(Cc), N + pn, M, N, K, epi, _lib.ptr(res),
                                              N + pn, _lib.ptr(gate), N, 3600, _lib.stream_ptr()))
        ctr = torch.zeros(8, dtype=torch.int32, device=dev)
        fn_t = lambda: _lib.check(lib.mmpl_gemm_tickets(_lib.ptr(A), K + pk, _lib.ptr(W), K + pk, _lib.ptr(b), _lib.ptr(Cc), N + pn, M, N, K, epi, _lib.ptr(res),
                                                        N + pn, _lib.ptr(gate), N, 3600, _lib.ptr(ctr), _lib.stream_ptr()))
        nb = lib.mmpl_gemm_scratch_bytes()
        scratch = torch.zeros(nb, dtype=torch.uint8, device=dev)
        fn_s = lambda: _lib.check(lib.mmpl_gemm_scratch(_lib.ptr(A), K + pk, _lib.ptr(W), K + pk, _lib.ptr(b), _lib.ptr(Cc), N + pn, M, N, K, epi, _lib.ptr(res),
                                                        N + pn, _lib.ptr(gate), N, 3600, _lib.ptr(scratch), nb, _lib.stream_ptr()))
        ms, ms_t, ms_s = timeit(fn, iters), timeit(fn_t, iters), timeit(fn_s, iters)
        ms2, ms_t2, ms_s2 = timeit(fn, iters), timeit(fn_t, iters), timeit(fn_s, iters)
        ms, ms_t, ms_s = min(ms, ms2), min(ms_t, ms_t2), min(ms_s, ms_s2)
        print(f"gemm {name}: M={M} N={N} K={K} epi={epi}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s   |  tile tickets {ms_t:8.3f} ms  "
              f"{2.0 * M * N * K / ms_t / 1e9:8.1f} TFLOP/s   |  + split-K tail {ms_s:8.3f} ms  {2.0 * M * N * K / ms_s / 1e9:8.1f} TFLOP/s", flush=True)


def bench_gemm_ref(iters):
    """yardstick only (never on the product path): the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the same shapes
    and the same random operands, plain C = A W^T without bias / epilogue"""
    shapes = (("qkv", 25200, 15360, 5120), ("o", 25200, 5120, 5120), ("ffn0", 25200, 13824, 5120), ("ffn2", 25200, 5120, 13824),
              ("qkv_s0", 7200, 15360, 5120), ("sq8k", 8192, 8192, 8192))
    if os.environ.get("BENCH_SHAPES"):
        shapes = tuple((f[0], int(f[1]), int(f[2]), int(f[3])) for f in (x.split(":") for x in os.environ["BENCH_SHAPES"].split(",")))
    for name, M, N, K in shapes:
        A = torch.randn(M, K, device=dev).to(BF)
        W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
        out = torch.empty(M, N, device=dev, dtype=BF)
        fn = lambda: torch.matmul(A, W.t(), out=out)
        ms = min(timeit(fn, iters), timeit(fn, iters))
        print(f"vendor gemm {name}: M={M} N={N} K={K}  {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TFLOP/s", flush=True)


def gemm_phases(epi=3):
    """needs a -DMMPL_DEV_ABLATIONS -DGEMM6_TIMING=1 build (tools/r06_gpu.sh gemmphases): per-wave { prologue, k loop, epilogue } cycles written over the output"""
    M, N, K = (int(v) for v in os.environ.get("BENCH_PHASE_SHAPE", "25200:5120:5120").split(":"))
    A = torch.randn(M, K, device=dev).to(BF)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
    b = torch.randn(N, device=dev).to(BF)
    Cc = torch.zeros(M, N, device=dev, dtype=BF)
    res = torch.randn(M, N, device=dev).to(BF)
    gate = torch.randn(8, N, device=dev).to(BF)
    for _ in range(2):
        _lib.check(lib.mmpl_gemm(_lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(b), _lib.ptr(Cc), N, M, N, K, epi, _lib.ptr(res), N, _lib.ptr(gate), N,
                                 3600, _lib.stream_ptr()))
    torch.cuda.synchronize()
    nblk = ((M + 255) // 256) * (N // 256)
    t = Cc.view(-1).view(torch.float32)[: nblk * 16].view(-1, 4).double()
    t = t[t[:, 3] == K // 32]
    print(f"gemmphases {M}x{N}x{K} epi={epi}: waves {len(t)}  prologue {t[:, 0].mean().item():.0f}  loop {t[:, 1].mean().item():.0f} "
          f"({t[:, 1].mean().item() / (K // 32) / 32:.2f} per MFMA; p5 {t[:, 1].quantile(0.05).item():.0f} p50 {t[:, 1].quantile(0.5).item():.0f} "
          f"p95 {t[:, 1].quantile(0.95).item():.0f})  epilogue {t[:, 2].mean().item():.0f} (max {t[:, 2].max().item():.0f}) cycles", flush=True)


def bench_elem(iters):
    """the HBM-bound passes of a block at the stage shapes: LayerNorm + modulation (3 per block) and q/k RMS-norm + RoPE + K page write
    (1 per block; the cross-attention's q norm is the same kernel on q alone).  Prints time, algorithmic GB/s and a hash of the outputs."""
    import hashlib
    from mmpl_amd.dit import DitEngine

    def sha(*ts):
        h = hashlib.sha256()
        for t in ts:
            h.update(t.contiguous().view(torch.int16).cpu().numpy().tobytes())
        return h.hexdigest()[:12]

    for model, d, H, lat, S, frames_list in (("14B/720p", 5120, 40, (90, 160), 3600, (7, 6, 2)), ("14B/480p", 5120, 40, (60, 104), 1560, (7, 6, 2)),
                                             ("1.3B/480p", 1536, 12, (60, 104), 1560, (7, 6, 2))):
        eng = DitEngine(dict(dim=d, ffn_dim=256, num_heads=H, num_layers=1, text_dim=64), lat[0], lat[1], dev)
        for nF in frames_list:
            rows = nF * S
            torch.manual_seed(nF)
            x = (torch.randn(rows, d, device=dev) * 3 + 0.5).to(BF)
            e = (torch.randn(nF, 6, d, device=dev) * 0.3).to(BF)
            y = torch.empty_like(x)
            fn = lambda: _lib.check(lib.mmpl_layernorm(_lib.ptr(x), d, _lib.ptr(y), d, rows, d, 1e-6, _lib.ptr(e[:, 1]), _lib.ptr(e[:, 0]), 6 * d, S,
                                                       None, None, _lib.stream_ptr()))
            ms = min(timeit(fn, iters), timeit(fn, iters))
            gb = 2.0 * rows * d * 2 / 1e9
            h = sha(y)
            w = (1 + 0.1 * torch.randn(d, device=dev)).to(BF)
            b = (0.1 * torch.randn(d, device=dev)).to(BF)
            _lib.check(lib.mmpl_layernorm(_lib.ptr(x), d, _lib.ptr(y), d, rows, d, 1e-6, None, None, 0, S, _lib.ptr(w), _lib.ptr(b), _lib.stream_ptr()))
            print(f"elem {model} layernorm+mod rows={rows} d={d}  {ms * 1e3:8.1f} us  {gb / ms:6.2f} TB/s  sha {h} affine {sha(y)}", flush=True)
            qkv0 = torch.randn(rows, 3 * d, device=dev).to(BF)
            qkv = qkv0.clone()
            wq = (1 + 0.1 * torch.randn(d, device=dev)).to(BF)
            wk = (1 + 0.1 * torch.randn(d, device=dev)).to(BF)
            kc = torch.zeros(nF * S, d, device=dev, dtype=BF)
            kd = (C.c_void_p * nF)(*[kc[i * S:].data_ptr() for i in range(nF)])
            fi = (C.c_int * nF)(*range(3, 3 + nF))
            # v = NULL as in the forward (the QKV GEMM's epilogue writes the V pages)
            fq = lambda: _lib.check(lib.mmpl_qknorm_rope(eng._h, _lib.ptr(qkv), 3 * d, _lib.ptr(qkv[:, d:]), 3 * d, None, 3 * d, _lib.ptr(wq), _lib.ptr(wk),
                                                         nF, fi, kd, kd, _lib.stream_ptr()))
            fq()
            torch.cuda.synchronize()
            h = sha(qkv[:, :d], kc)
            ms = min(timeit(fq, iters), timeit(fq, iters))       # q is normalised in place again and again: same work, values irrelevant
            gb = 4.0 * rows * d * 2 / 1e9
            print(f"elem {model} qknorm+rope+kwrite rows={rows} d={d}  {ms * 1e3:8.1f} us  {gb / ms:6.2f} TB/s  sha {h}", flush=True)
            del qkv, qkv0, kc, x, y


def bench_cross(iters):
    """text cross-attention alone (attn_fwd_kernel<1>): Lq query rows x 40 heads against `keys` context rows (valid tokens + the one
    collapsed padding row), q and o streams of Lq x 5120 bf16 each; time, algorithmic TB/s of q + o, a hash of the output"""
    import hashlib
    for H, d, Lq, keys in [(40, 5120, Lq, keys) for Lq in (25200, 21600, 7200) for keys in (128, 100, 65, 41, 13)] + \
                          [(12, 1536, Lq, keys) for Lq in (10920, 9360, 3120) for keys in (65, 41)]:
        if True:
            torch.manual_seed(keys)
            q = torch.randn(Lq, d, device=dev).to(BF)
            k = torch.randn(512, d, device=dev).to(BF)
            v = torch.randn(512, d, device=dev).to(BF)
            o = torch.empty_like(q)
            kp, vp = (C.c_void_p * 1)(k.data_ptr()), (C.c_void_p * 1)(v.data_ptr())
            fn = lambda: _lib.check(lib.mmpl_attn_fwd_variant(_lib.ptr(q), d, _lib.ptr(o), d, kp, vp, d, d, 1, keys, Lq, H, 1.0 / math.sqrt(128.0),
                                                              None, 0, 0, 1, _lib.stream_ptr()))
            ms = min(timeit(fn, iters), timeit(fn, iters))
            h = hashlib.sha256(o.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
            print(f"cross H={H} Lq={Lq} keys={keys}  {ms * 1e3:8.1f} us  {2.0 * Lq * d * 2 / 1e9 / ms:6.2f} TB/s (q + o)  "
                  f"{4.0 * Lq * keys * d / ms / 1e9:7.1f} TFLOP/s  sha {h}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 5
    if what in ("attn", "all"):
        bench_attn(iters)
    if what == "gemmphases":
        gemm_phases(3)
        gemm_phases(0)
    if what in ("gemm", "all"):
        bench_gemm(iters)
    if what == "gemmref":
        bench_gemm_ref(iters)
    if what == "attnref":
        bench_attn_ref(iters)
    if what == "elem":
        bench_elem(iters)
    if what == "cross":
        bench_cross(iters)
