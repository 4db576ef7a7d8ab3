"""Stage schedule, KV-slot mapping and visibility bookkeeping of MMPL's "planning" denoise order.

Reproduces, as data, what the reference spreads over literals:
  * stage schedules:  T2V clean_steps (MMPL_t2v/pipeline/casual_fps_inference.py:250-252),
                      I2V clean_steps (MMPL_i2v/pipeline/casual_fps_inference.py:253-255)
  * slot rule:        frames 19,20 live in cache slots 13,14; the stage containing frame 15 never writes and
                      attends to its own K/V appended after the cache (wan/modules/causal_fps_model.py:209-264)
  * visibility edits: T2V hides frames 19,20 during the [4..9] stage and re-adds them for [13..18]
                      (casual_fps_inference.py:298-302, 321-325)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence

T2V_CLEAN_STEPS = [0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 1, 1, 1, 3, 3, 3, 3, 3, 3, 1, 1]
I2V_CLEAN_STEPS = [0, 1, 2, 2, 3, 3, 3, 3, 3, 3, 2, 2, 2, 4, 4, 4, 4, 4, 4, 2, 2]
HIDDEN_FRAMES = (20, 19)
N_SLOTS = 15


def slot_of(frame: int) -> int:
    return frame - 6 if frame >= 19 else frame


@dataclass
class StagePlan:
    mode: str = "t2v"
    stages: List[List[int]] = field(init=False)

    def __post_init__(self):
        clean = T2V_CLEAN_STEPS if self.mode == "t2v" else I2V_CLEAN_STEPS
        self.stages = [[i for i, v in enumerate(clean) if v == t] for t in range(max(clean) + 1)]

    @property
    def handoff_stage(self) -> int:
        return 1 if self.mode == "t2v" else 2

    @staticmethod
    def write_slots(frames: Sequence[int]) -> List[int]:
        if 15 in frames:
            return [-1] * len(frames)
        return [slot_of(f) for f in frames]

    def hides_anchors(self, stage_index: int) -> bool:
        return self.mode == "t2v" and stage_index == 2

    def shows_anchors(self, stage_index: int) -> bool:
        return self.mode == "t2v" and stage_index == 3

    def renoised_frames(self, stage_index: int):
        """T2V re-draws the first and last frame of the two in-fill stages from fresh noise
        (casual_fps_inference.py:284-294, 307-317; add_noise with t >= 1000 returns the noise itself)."""
        if self.mode == "t2v" and stage_index in (2, 3):
            fr = self.stages[stage_index]
            return [fr[0], fr[-1]]
        return []


class VisibleFrames:
    """`attention_vis_index` as frame ids (the reference stores token offsets frame*1560)."""

    def __init__(self):
        self.frames: List[int] = []

    def on_forward(self, frames: Sequence[int]) -> None:
        if 15 not in frames:
            for f in frames:
                if f not in self.frames:
                    self.frames.append(f)

    def hide(self, frames=HIDDEN_FRAMES) -> None:
        for f in frames:
            if f in self.frames:
                self.frames.remove(f)

    def show(self, frames=HIDDEN_FRAMES) -> None:
        for f in frames:
            if f not in self.frames:
                self.frames.append(f)

    def slots(self) -> List[int]:
        return [slot_of(f) for f in self.frames]

    def token_offsets(self, frame_seqlen: int) -> List[int]:
        return [f * frame_seqlen for f in self.frames]


def dit_forward_flops(cfg: dict, frame_seqlen: int, n_q_frames: int, n_kv_frames: int, text_len: int = 512) -> float:
    """Algorithmic FLOPs of one DiT forward, SURVEY.md 8(d):  L*[2*Lq*(6d^2 + 2df) + 4*Lq*Lkv*d + 4*Lq*512*d]."""
    d, f, L = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    Lq, Lkv = n_q_frames * frame_seqlen, n_kv_frames * frame_seqlen
    return L * (2.0 * Lq * (6.0 * d * d + 2.0 * d * f) + 4.0 * Lq * Lkv * d + 4.0 * Lq * text_len * d)


def dit_forward_flops_executed(cfg: dict, frame_seqlen: int, n_q_frames: int, n_kv_frames: int, cross_keys: int,
                               block0_self_attn_shared: bool = False) -> float:
    """FLOPs of the launches a forward actually issues, where they differ from the reference's algorithmic count
    (`dit_forward_flops`): the text cross-attention over `cross_keys` keys (the padded tail collapsed: CrossKV.rows + 1 instead of
    512) and, for the uncond forward of a step whose branches run back to back, no block-0 self-attention and no block-0 output
    projection (`share_in`).  Both are bit-exact savings; the algorithmic count is what `roofline` and SURVEY.md 8(d) price."""
    d, f, L = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    Lq, Lkv = n_q_frames * frame_seqlen, n_kv_frames * frame_seqlen
    fl = L * (2.0 * Lq * (6.0 * d * d + 2.0 * d * f) + 4.0 * Lq * Lkv * d + 4.0 * Lq * cross_keys * d)
    if block0_self_attn_shared:
        fl -= 4.0 * Lq * Lkv * d + 2.0 * Lq * d * d
    return fl


# (query frames, attended frames) per T2V stage, first chunk
T2V_STAGE_SHAPES = [(2, 2), (7, 9), (6, 13), (6, 21)]
# I2V denoise stages s1..s4 (s0 = the image latent, refresh pass only; frames 19, 20 stay visible in s3): SURVEY.md App. A
I2V_STAGE_SHAPES = [(1, 2), (7, 9), (6, 15), (6, 21)]


def assemble_chunk_seconds(step_seconds, first_step_index: int, stage_flops, steps_per_stage: int = 51):
    """bench.py: K timed steps rotate through the four stage shapes (step i runs stage (first_step_index + i) % 4).
    Returns (per-stage mean step time, chunk time = steps_per_stage * sum of them).  A stage the K steps never reached is
    priced at the FLOP rate measured on the others, so any K >= 1 gives an unbiased estimate."""
    per_stage = [[t for i, t in enumerate(step_seconds) if (first_step_index + i) % 4 == k] for k in range(4)]
    seen = [k for k in range(4) if per_stage[k]]
    if not seen:
        raise ValueError("no timed steps")
    rate = sum(stage_flops[k] for k in seen) / sum(sum(per_stage[k]) / len(per_stage[k]) for k in seen)
    stage_s = [sum(per_stage[k]) / len(per_stage[k]) if per_stage[k] else stage_flops[k] / rate for k in range(4)]
    return stage_s, float(steps_per_stage) * sum(stage_s)


# Measured on MI355X (profiles/r05c_bench_concurrent_cfg_ab.log, r05d_bench_concurrent_cfg_rule.log): with the two CFG branches of a
# denoise step captured as PARALLEL branches of the step graph, the tails of one branch's kernels (partial last rounds of tiles,
# split-KV tails, launch gaps, the chip-wide prologue / epilogue phases of short-K GEMMs) are filled by the other's.  By query rows x
# model dim of the stage: Wan 1.3B at 480p (4.8 ... 16.8 M) -5.6 ... -17 % per stage, -8.4 ... -9.9 % per step; Wan 14B at 480p 16 M:
# -12 %, 55.9 M: -1.7 %, 47.9 M: 0 ... -0.3 %; Wan 14B at 720p 36.9 M: -1.3 ... -2.4 %, 111 ... 129 M: -0.5 ... +2 % by box (two
# chip-filling kernels contending for the same CUs and L2 lose about what their tails are worth).
CONCURRENT_CFG_MAX_ROWS_X_DIM = 60e6


def concurrent_cfg_pays(n_query_rows: int, dim: int) -> bool:
    """Should the cond and the uncond forward of a denoise step run as parallel graph branches?  (bit-identical either way)"""
    return float(n_query_rows) * float(dim) <= CONCURRENT_CFG_MAX_ROWS_X_DIM
