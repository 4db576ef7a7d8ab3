"""Shared helpers for the parity tests."""
import os

import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def max_abs(a: torch.Tensor, b: torch.Tensor) -> float:
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def bf16_ulp_frac(a: torch.Tensor, b: torch.Tensor, ulps: int = 1) -> float:
    """fraction of elements that differ by more than `ulps` bf16 ulps (compared on the raw bit patterns)."""
    ai = a.cpu().contiguous().view(torch.int16).to(torch.int32)
    bi = b.cpu().contiguous().view(torch.int16).to(torch.int32)
    # map sign-magnitude to a monotone integer line
    ai = torch.where(ai < 0, -(ai & 0x7fff), ai)
    bi = torch.where(bi < 0, -(bi & 0x7fff), bi)
    return ((ai - bi).abs() > ulps).float().mean().item()
