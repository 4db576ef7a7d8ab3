#!/usr/bin/env python3
"""The coefficients of attn_w64.hip's exp2 polynomial: minimax RELATIVE error fit of 2^f on [0, 1) (Lawson's iteratively reweighted
least squares on a dense grid), and what the approximation does to an attention output next to the bf16 rounding of P.
    python tools/fit_exp2_poly.py [degree]        # default 3: max relative error 7.49e-5 = 2^-13.7"""
import sys

import numpy as np


def fit(deg, n=20001, iters=200):
    x = np.linspace(0.0, 1.0, n)
    y = 2.0 ** x
    w = np.ones_like(x)
    A = np.vander(x, deg + 1, increasing=True) / y[:, None]          # relative error: p(x) / y - 1
    for _ in range(iters):
        sw = np.sqrt(w)
        c, *_ = np.linalg.lstsq(A * sw[:, None], sw, rcond=None)
        r = np.abs(A @ c - 1.0)
        w = w * (r / r.max() + 1e-12)
        w /= w.sum()
    return c, r.max()


def bf16(a):
    u = np.asarray(a, dtype=np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7FFF)) & 0xFFFF0000).view(np.float32)


if __name__ == "__main__":
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    c, e = fit(deg)
    print(f"degree {deg}: c0..c{deg} =", ", ".join(repr(float(v)) for v in c), f"  max relative error {e:.3e} = 2^{np.log2(e):.1f}")
    c32 = c.astype(np.float32)
    rng = np.random.default_rng(0)
    for sigma in (1.0, 4.0, 16.0):                                    # logit spread in nats
        s = (rng.standard_normal((64, 4096)) * sigma * 1.4427).astype(np.float32)
        s = s - s.max(1, keepdims=True) - 64.0
        v = rng.standard_normal((4096, 128)).astype(np.float32)
        ex = np.exp2(s.astype(np.float64))
        ref = (ex @ v) / ex.sum(1, keepdims=True)
        pe = np.exp2(s).astype(np.float32)
        oe = (bf16(pe).astype(np.float64) @ v) / pe.astype(np.float64).sum(1, keepdims=True)
        n = np.floor(s)
        f = (s - n).astype(np.float32)
        q = np.zeros_like(f)
        for k in range(deg, -1, -1):
            q = np.float32(q * f + c32[k])
        pa = np.ldexp(q, n.astype(np.int32)).astype(np.float32)
        oa = (bf16(pa).astype(np.float64) @ v) / pa.astype(np.float64).sum(1, keepdims=True)
        rl = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)   # noqa: E731
        print(f"  sigma {sigma:4.1f} nats: rel-L2 vs fp64 of O with bf16 P -- exact exp2 {rl(oe, ref):.2e}, polynomial {rl(oa, ref):.2e}; polynomial vs exact {rl(oa, oe):.2e}")
