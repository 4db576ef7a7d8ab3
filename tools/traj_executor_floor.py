#!/usr/bin/env python3
"""dev / evidence: the trajectory-level distance between two EXECUTORS of the reference's own algorithm.  The 408-forward fixture
(tests/golden/chunk_t2v_tiny_50.pt) was produced by the reference on a CPU; the reference's native platform is a GPU.  This runs the
oracle's stage loop -- the reference's PyTorch ops, restated -- with every tensor on the device (rocBLAS / PyTorch SDPA kernels) and
prints its distance to the fixture: what "the reference run on another platform" costs after 50 steps x CFG 5 x 4 stages, i.e. the
floor any GPU implementation, the reference's own included, can be held to.  Test infrastructure only (imports oracle/)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal  # noqa: E402
from oracle import stage_ref  # noqa: E402
from oracle import wan_dit_ref as W  # noqa: E402
from tests.util import GOLDEN, rel_l2  # noqa: E402

H, Wd = 60, 104
fx = torch.load(f"{GOLDEN}/chunk_t2v_tiny_50.pt")
m, nf = fx["meta"], fx["noise_floor"]
cfg = WAN_CONFIGS[m["cfg"]]
dev = "cuda:0"
sd = {k: v.to(dev) for k, v in dit_state_dict(cfg, seed=m["weight_seed"]).items()}
ctxs = []
for seed, nv in zip(m["ctx_seeds"], m["n_valid"]):
    c = philox_normal([512, cfg["text_dim"]], seed)
    c[nv:] = 0
    ctxs.append(c.to(dev))
noise = philox_normal([1, 21, 16, H, Wd], m["noise_seed"]).to(dev)
renoise = {f: philox_normal([1, 16, H, Wd], m["renoise_seed_base"] + f).to(dev) for f in (4, 9, 13, 18)}
t0 = time.time()
with torch.device(dev):
    out, hand, _ = stage_ref.run_chunk(sd, W.DitCfg(**cfg), noise, ctxs[0], ctxs[1], renoise, None, "t2v", m["guidance"], m["steps"], m["shift"])
torch.cuda.synchronize()
out, hand = out.cpu(), hand.cpu()
print(f"oracle ops executed ON THE DEVICE vs the reference's CPU run, 408 forwards at 60x104 ({time.time() - t0:.0f} s): "
      f"final latents {rel_l2(out[..., ::2, ::2], fx['out_strided']):.3e}, hand-off {rel_l2(hand[..., ::3, ::3], fx['handoff_strided']):.3e}, "
      f"vs the reference's fp32 run {rel_l2(out[..., ::2, ::2], fx['out_f32_strided']):.3e}  "
      f"(reference vs itself, K/V order: {nf['order_out']:.3e}; reference bf16 vs fp32: {nf['f32_out']:.3e})")
