#!/usr/bin/env python3
"""Dev tool: parity + same-process A/B of the attention kernel variants behind mmpl_attn_fwd_variant.

    python tools/attn_dev.py check [variants...]     # edge-case shapes vs an fp32 restatement evaluated by torch on the device
    python tools/attn_dev.py bench [variants...]     # 14B/720p stage shapes, interleaved rounds, TFLOP/s per variant
variants: 1 lock-step, 3 w64 on a raw q, 4 w64 on a producer-prescaled q (default: 2 4)
"""
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
WS = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev)


def run(variant, q, ldq, o, d, kp, vp, n_pages, S, Lq, H, ws=True):
    _lib.check(lib.mmpl_attn_fwd_variant(_lib.ptr(q), ldq, _lib.ptr(o), d, kp, vp, d, d, n_pages, S, Lq, H, 1.0 / math.sqrt(128),
                                         _lib.ptr(WS) if ws else None, WS.numel() if ws else 0, variant, 0, _lib.stream_ptr()))


CQ = (1.0 / math.sqrt(128)) * 1.4426950408889634


def ref_fp32(q, k, v, H):
    Lq, Lk = q.shape[0], k.shape[0]
    qq = q.float().view(Lq, H, 128).transpose(0, 1)
    kk = k.float().view(Lk, H, 128).transpose(0, 1)
    vv = v.float().view(Lk, H, 128).transpose(0, 1)
    s = torch.softmax(qq @ kk.transpose(1, 2) / math.sqrt(128), dim=-1)
    return (s @ vv).transpose(0, 1).reshape(Lq, H * 128)


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def check(variants):
    cases = [(96, 2, 96, 1), (200, 2, 100, 3), (512, 1, 512, 1), (3120, 2, 1560, 2), (300, 8, 72, 21), (257, 3, 40, 5),
             (64, 1, 64, 1), (130, 2, 30, 1), (100, 1, 64, 3), (70, 2, 128, 2), (256, 1, 10, 24), (256 * 41 - 57, 8, 640, 3),
             (7200, 8, 3600, 2)]
    bad = 0
    for Lq, H, S, n_pages in cases:
        torch.manual_seed(Lq + S)
        d = H * 128
        ldm = 3 if H == 2 else 1
        q32 = torch.randn(Lq, ldm * d, device=dev)
        q = q32.to(BF)
        qp = (q32 * CQ).to(BF)                       # variant 4: scale folded in before the rounding
        n_slots = n_pages + 2
        kc = torch.randn(n_slots * S, d, device=dev).to(BF)
        vc = torch.randn(n_slots * S, d, device=dev).to(BF)
        slots = torch.randperm(n_slots)[:n_pages].tolist()
        kp = (C.c_void_p * n_pages)(*[kc[s * S:].data_ptr() for s in slots])
        vp = (C.c_void_p * n_pages)(*[vc[s * S:].data_ptr() for s in slots])
        idx = torch.cat([torch.arange(s * S, (s + 1) * S) for s in slots]).to(dev)
        refs = {False: ref_fp32(q[:, :d], kc[idx], vc[idx], H), True: ref_fp32(qp[:, :d].float() / CQ, kc[idx], vc[idx], H)}
        line = f"Lq={Lq} H={H} S={S} pages={n_pages}:"
        for var in variants:
            o = torch.full((Lq, d), float("nan"), device=dev, dtype=BF)
            run(var, qp if var == 4 else q, ldm * d, o, d, kp, vp, n_pages, S, Lq, H)
            torch.cuda.synchronize()
            ref = refs[var == 4]
            e = rel_l2(o, ref) if torch.isfinite(o.float()).all() else float("nan")
            line += f"  v{var} {e:.2e}"
            if not (e < 1e-2):
                bad += 1
                line += " <-- BAD"
        print(line, flush=True)
    # spiked scores: the running reference must move late in the stream
    torch.manual_seed(3)
    Lq, S = 128, 320
    q = torch.randn(Lq, 128, device=dev).to(BF)
    k = torch.randn(S, 128, device=dev).to(BF)
    v = torch.randn(S, 128, device=dev).to(BF)
    k[300] = (q[5].float() * 4).to(BF)
    k[70] = (q[17].float() * 3).to(BF)
    k[200] = (q[40].float() * 30).to(BF)          # far beyond any deferral threshold
    kp = (C.c_void_p * 1)(k.data_ptr())
    vp = (C.c_void_p * 1)(v.data_ptr())
    qp = (q.float() * CQ).to(BF)
    for var in variants:
        o = torch.full((Lq, 128), float("nan"), device=dev, dtype=BF)
        run(var, qp if var == 4 else q, 128, o, 128, kp, vp, 1, S, Lq, 1)
        torch.cuda.synchronize()
        ref = ref_fp32(qp.float() / CQ if var == 4 else q, k, v, 1)
        e, m = rel_l2(o, ref), (o.float() - ref).abs().max().item()
        tol = 2.0 ** -7 * ref.abs().max().item()       # 2 bf16 ulps of the largest output
        print(f"spiked: v{var} rel {e:.2e} max {m:.2e}" + ("" if e < 1e-2 and m < tol else " <-- BAD"), flush=True)
        bad += not (e < 1e-2 and m < tol)
    print("CHECK", "FAILED" if bad else "OK", flush=True)
    return bad


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


def bench(variants, iters=3, rounds=3, stages=("s0", "s1", "s2", "s3")):
    H, S, d = 40, 3600, 5120
    kc = torch.randn(21 * S, d, device=dev).to(BF)
    vc = torch.randn(21 * S, d, device=dev).to(BF)
    for name, nq, npg in (("s0", 2, 2), ("s1", 7, 9), ("s2", 6, 13), ("s3", 6, 21)):
        if name not in stages:
            continue
        Lq = nq * S
        q32 = torch.randn(Lq, 3 * d, device=dev)
        q, qp = q32.to(BF), (q32 * CQ).to(BF)
        del q32
        kp = (C.c_void_p * npg)(*[kc[i * S:].data_ptr() for i in range(npg)])
        vp = (C.c_void_p * npg)(*[vc[i * S:].data_ptr() for i in range(npg)])
        outs, best = {}, {v: 0.0 for v in variants}
        line = f"attn {name}: Lq={Lq} Lkv={npg * S}"
        for _ in range(rounds):
            for var in variants:
                o = torch.zeros(Lq, d, device=dev, dtype=BF)
                ms = timeit(lambda: run(var, qp if var == 4 else q, 3 * d, o, d, kp, vp, npg, S, Lq, H), iters)
                outs[var] = o
                tf = 4.0 * Lq * npg * S * d / ms / 1e9
                best[var] = max(best[var], tf)
                line += f"  v{var} {tf:7.1f}"
        v0 = variants[0]
        diffs = "  rel-diff vs v%d: " % v0 + " ".join(f"v{v} {rel_l2(outs[v], outs[v0]):.1e}" for v in variants[1:])
        print(line + "  | best " + " ".join(f"v{v} {best[v]:.1f}" for v in variants) + diffs, flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    variants = [int(x) for x in sys.argv[2:] if x.isdigit()] or [1, 4]
    stages = [x.split("=")[1].split(",") for x in sys.argv[2:] if x.startswith("stages=")]
    if what == "check":
        sys.exit(1 if check(variants) else 0)
    bench(variants, stages=stages[0] if stages else ("s0", "s1", "s2", "s3"))
