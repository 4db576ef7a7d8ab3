"""TEST INFRASTRUCTURE ONLY -- a CPU restatement of the state byte of the self-attention kernel's pass history
(mmpl_amd/csrc/attn_w64.hip, include/mmpl_hip.h `attn_history`).  The reference has no counterpart: its attention is stateless
(wan/modules/attention.py:139-185); the history only decides WHICH of the kernel's two exact softmax passes runs.

State byte of one (head, 256-row query block, split part):
    bit 7      the 128 lane references next to it are valid (set at the block's first failure, never cleared until the caller zeroes)
    bits 5-6   back-off level (0..2)
    bits 0-4   countdown: > 1 -> the block goes straight to the GENERAL pass and the countdown is decremented;
               1 -> FAST is tried again; 0 -> FAST
"""
MEM = 128


def plan(state: int):
    """(try_fast, use_remembered_references) of a launch that finds `state`."""
    return (state & 31) <= 1, bool(state & MEM)


def next_state(state: int, fast_failed: bool) -> int:
    """The byte the launch leaves.  `fast_failed`: did the FAST pass (if it ran) leave a row sum outside the window?"""
    try_fast, have_mem = plan(state)
    if not try_fast:
        return state - 1
    if not fast_failed:
        return state & MEM
    if not have_mem:
        return MEM                                     # first failure: references stored, FAST again next time (on them)
    level = min(((state >> 5) & 3) + 1, 2) if (state & 31) else 0
    return MEM | (level << 5) | min(8 << level, 31)


def passes_paid(state: int, fast_failed: bool) -> float:
    """Cost of the launch in FAST-pass times (the GENERAL pass is ~1.66, profiles/NOTEBOOK_r04.md section C)."""
    try_fast, _ = plan(state)
    if not try_fast:
        return 1.66
    return 2.66 if fast_failed else 1.0
