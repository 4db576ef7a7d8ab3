#!/usr/bin/env python3
"""Measured (not extrapolated) wall-clock of ONE complete chunk through the real pipeline (dev/evidence tool):
4 T2V stages x 50 UniPC steps x CFG + refresh passes (+ VAE decode), synthetic weights.
    python tools/full_chunk.py --model 14B --res 720p"""
import argparse
import json
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmpl_amd.geometry import Geometry  # noqa: E402
from mmpl_amd.pipeline import CausalFPSInferencePipeline  # noqa: E402
from mmpl_amd.stage_plan import T2V_STAGE_SHAPES, dit_forward_flops  # noqa: E402
from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, vae_state_dict  # noqa: E402
from mmpl_amd.wan_wrapper import SyntheticTextEncoder, WanFPSWrapper, WanVAEWrapper  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="14B")
ap.add_argument("--res", default="720p")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--no-graphs", action="store_true")
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = "cuda:0"
cfg = WAN_CONFIGS[a.model]
geo = Geometry.named(a.res)
gen = WanFPSWrapper(is_causal=True, timestep_shift=5.0, model_config=cfg, geometry=geo, device=dev)
gen.load_state_dict(dit_state_dict(cfg, seed=1, device=dev))
vae = WanVAEWrapper(geometry=geo, device=dev, state_dict=vae_state_dict(seed=2))
args = types.SimpleNamespace(model_kwargs={}, num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0, negative_prompt="bad",
                             independent_first_frame=False, sampling_steps=a.steps)
pipe = CausalFPSInferencePipeline(args, dev, generator=gen, text_encoder=SyntheticTextEncoder(cfg["text_dim"], dev), vae=vae, save=None,
                                  geometry=geo)
pipe.use_graphs = not a.no_graphs
handoff = {}
pipe.handoff_sink = lambda t: handoff.setdefault("t", t.clone())
noise = torch.randn(1, 21, 16, geo.lat_h, geo.lat_w, device=dev).to(torch.bfloat16)
torch.cuda.synchronize()
t0 = time.perf_counter()
_, lat = pipe.inference(noise, ["a cat"], return_latents=True, decode=False)
torch.cuda.synchronize()
t1 = time.perf_counter()
video = pipe.vae.decode_to_pixel(lat)
torch.cuda.synchronize()
t2 = time.perf_counter()
# consumer side of the inter-chunk hand-off (decode 4 latents -> 13 px frames -> encode 5 frames -> 2 latents)
from mmpl_amd.handoff import handoff_to_initial_latent  # noqa: E402
handoff_to_initial_latent(pipe.vae, handoff["t"])
torch.cuda.synchronize()
t3 = time.perf_counter()
init = handoff_to_initial_latent(pipe.vae, handoff["t"])
torch.cuda.synchronize()
t4 = time.perf_counter()
S = geo.frame_seqlen
fl = [dit_forward_flops(cfg, S, q, kv) for q, kv in T2V_STAGE_SHAPES]
n_fwd = [2 * a.steps + 2] * 3 + [2 * a.steps]          # the non-persisting stage skips its no-op refresh pair
flops = sum(n * f for n, f in zip(n_fwd, fl))
print(json.dumps({"model": a.model, "res": a.res, "sampling_steps": a.steps, "hipgraphs": pipe.use_graphs,
                  "denoise_s": t1 - t0, "vae_decode_s": t2 - t1, "handoff_transform_s": t4 - t3,
                  "handoff_shape": list(handoff["t"].shape), "initial_latent_shape": list(init.shape), "latent_frames_per_s_denoise": 21 / (t1 - t0),
                  "latent_frames_per_s_end_to_end": 21 / (t2 - t0), "dit_forwards": sum(n_fwd), "algorithmic_pflop": flops / 1e15,
                  "achieved_pflops": flops / (t1 - t0) / 1e15, "finite": bool(torch.isfinite(video).all()),
                  "video_shape": list(video.shape)}))
