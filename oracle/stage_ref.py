"""Oracle: the FPS stage loop of ``CausalFPSInferencePipeline.inference`` re-enacted on explicit tensors.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  * T2V: MMPL_t2v/pipeline/casual_fps_inference.py:250-439 (stage schedule, re-noise of frames 4/9 and
    13/18, hiding / re-adding frames 19,20, CFG loop :338-374, hand-off :380-383, refresh pass :385-403,
    initial-latent path :407-439)
  * I2V: MMPL_i2v/pipeline/casual_fps_inference.py:253-435 (schedule [0],[1],anchors,[4..9],[13..18];
    hand-off after the anchor stage :340-343; no re-noise, no hiding)
  * slot / visibility rule: MMPL_t2v/wan/modules/causal_fps_model.py:209-264
  * hand-off consumer: MMPL_t2v/Wan_fps_inference_parallel_4gpu_20s.py:191-205

RNG is never drawn here: every random tensor the reference draws (the chunk noise and the re-drawn
frames) is an explicit input, see SURVEY.md section 8(c).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import wan_dit_ref as W
from .unipc_ref import FlowUniPCRef

T2V_CLEAN_STEPS = [0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 1, 1, 1, 3, 3, 3, 3, 3, 3, 1, 1]     # t2v :250
I2V_CLEAN_STEPS = [0, 1, 2, 2, 3, 3, 3, 3, 3, 3, 2, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2]     # i2v :253
HIDDEN_FRAMES = (20, 19)                                                                 # 31200, 29640 (/1560)
N_SLOTS = 15                                                                             # (32760 - 6*1560) / 1560


def stage_frames(clean_steps: Sequence[int]) -> List[List[int]]:
    return [[i for i, v in enumerate(clean_steps) if v == t] for t in range(max(clean_steps) + 1)]


def slot_of(frame: int) -> int:
    """causal_fps_model.py:220,234-236: frames 19,20 live in slots 13,14."""
    return frame - 6 if frame >= 19 else frame


def write_slots_for(frames: Sequence[int]) -> List[int]:
    """causal_fps_model.py:209-241: the stage containing frame 15 never writes."""
    if 15 in frames:
        return [-1] * len(frames)
    return [slot_of(f) for f in frames]


class VisIndex:
    """`attention_vis_index` bookkeeping (frame ids instead of token offsets)."""

    def __init__(self):
        self.frames: List[int] = []

    def on_forward(self, frames: Sequence[int]):
        if 15 not in frames:                                   # :219 / :243 vs :255
            for f in frames:
                if f not in self.frames:
                    self.frames.append(f)

    def hide(self, frames=HIDDEN_FRAMES):                      # pipeline :298-302
        for f in frames:
            if f in self.frames:
                self.frames.remove(f)

    def show(self, frames=HIDDEN_FRAMES):                      # pipeline :321-325
        for f in frames:
            if f not in self.frames:
                self.frames.append(f)

    def slots(self) -> List[int]:
        return [slot_of(f) for f in self.frames]


def run_chunk(p: Dict[str, torch.Tensor], cfg: W.DitCfg, noise: torch.Tensor, ctx_cond: torch.Tensor,
              ctx_uncond: torch.Tensor, renoise: Optional[Dict[int, torch.Tensor]] = None,
              initial_latent: Optional[torch.Tensor] = None, mode: str = "t2v", guidance: float = 5.0,
              steps: int = 50, shift: float = 5.0, attn_fn=W.sdpa, trace: Optional[list] = None,
              gpu_scalar_semantics: bool = False, clip_fea: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
              last_stage: Optional[int] = None):
    """noise: [1, 21, 16, h, w]; renoise: {frame: [1,16,h,w]} replacements for frames 4,9,13,18 (t2v only);
    initial_latent: [1, 2, 16, h, w] or None.  Returns (output latents [1,21,16,h,w], hand-off tensor).
    clip_fea [257, clip_dim] and y [20, 21, h, w]: the Wan-I2V model type (p holds img_emb / k_img / v_img, cfg.in_dim 36) --
    every forward sees its frames of y concatenated to the latents on the channel axis (wan/modules/model.py:680-681) and
    both CFG branches share clip_fea and y (wan/image2video.py:283-295).
    last_stage: stop after that stage (tests that only need the hand-off: it exists once the anchor stage is done)."""
    clean = T2V_CLEAN_STEPS if mode == "t2v" else I2V_CLEAN_STEPS
    stages = stage_frames(clean)
    S = (noise.shape[-2] // 2) * (noise.shape[-1] // 2)
    caches = [W.new_kv_cache(cfg, N_SLOTS, S, noise.dtype) for _ in range(2)]
    cross = [[None] * cfg.num_layers for _ in range(2)]
    vis = [VisIndex(), VisIndex()]
    ctxs = [ctx_cond, ctx_uncond]
    output = torch.zeros_like(noise)
    handoff = None

    def fwd(which, lat, tval, frames):
        vis[which].on_forward(frames)
        t = torch.full([1, len(frames)], float(tval), dtype=torch.float32)
        x = lat[0].permute(1, 0, 2, 3)
        if y is not None:
            x = torch.cat([x, y[:, frames].to(x.dtype)], dim=0)
        o = W.dit_forward(p, cfg, x, t, ctxs[which], caches[which], cross[which], frames,
                          write_slots_for(frames), vis[which].slots(), attn_fn, clip_fea=clip_fea)
        return o.permute(1, 0, 2, 3).unsqueeze(0)

    def refresh(lat, frames):
        for which in (0, 1):
            fwd(which, lat, 0.0, frames)

    if mode == "t2v":
        first_denoised = 0 if initial_latent is None else 1
        if initial_latent is not None:                                           # :407-439
            refresh(initial_latent, stages[0])
            output[:, stages[0]] = initial_latent
    else:
        # i2v: stage 0 is the image latent; chunks >= 2 pass two frames and refresh both (i2v :369-435)
        assert initial_latent is not None
        n_init = initial_latent.shape[1]
        for j in range(n_init):
            refresh(initial_latent[:, j:j + 1], stages[j])
            output[:, stages[j]] = initial_latent[:, j:j + 1]
        first_denoised = n_init

    for si in range(first_denoised, len(stages)):
        frames = stages[si]
        latents = noise[:, frames].clone()
        if mode == "t2v" and si in (2, 3):
            if renoise is not None:
                latents[:, 0:1] = renoise[frames[0]].unsqueeze(1)              # add_noise(t>=1000) == fresh noise
                latents[:, -1:] = renoise[frames[-1]].unsqueeze(1)
            (vis[0].hide(), vis[1].hide()) if si == 2 else (vis[0].show(), vis[1].show())
        sched = FlowUniPCRef(1000, 2, 1.0, gpu_scalar_semantics)
        sched.set_timesteps(steps, shift=shift)
        for t in sched.timesteps:
            fc = fwd(0, latents, t.item(), frames)
            fu = fwd(1, latents, t.item(), frames)
            flow = fu + guidance * (fc - fu)                                      # :366-367 (python float: fp32 on CPU and GPU)
            latents = sched.step(flow, latents)
            if trace is not None:
                trace.append((si, int(t), flow.clone(), latents.clone()))
        output[:, frames] = latents
        handoff_stage = 1 if mode == "t2v" else 2
        if si == handoff_stage:
            handoff = (torch.cat([output[:, :1], latents], dim=1) if mode == "t2v"
                       else torch.cat([output[:, :1], output[:, -2:]], dim=1))
        refresh(latents, frames)                                                  # :385-403
        if last_stage is not None and si >= last_stage:
            break
    return output, handoff, caches


def handoff_to_mask_latents(recv: torch.Tensor) -> torch.Tensor:
    """Wan_fps_inference_parallel_4gpu_20s.py:191-195: [f0, f19, f19, f20, 0...] (21 frames)."""
    m = torch.zeros(1, 21, *recv.shape[2:], dtype=recv.dtype)
    m[:, 0] = recv[:, 0]
    m[:, 1] = recv[:, -2]
    m[:, 2:4] = recv[:, -2:]
    return m
