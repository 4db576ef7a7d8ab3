"""Host logic: stage schedule, slot map, visibility evolution, algorithmic-FLOP table (known answers from
SURVEY.md Appendix A / BASELINE.md section 3)."""
import pytest

from mmpl_amd.stage_plan import (T2V_STAGE_SHAPES, StagePlan, VisibleFrames, dit_forward_flops, slot_of)
from mmpl_amd.synthetic import WAN_CONFIGS
from oracle import stage_ref


def test_t2v_stage_tables():
    p = StagePlan("t2v")
    assert p.stages == [[0, 1], [2, 3, 10, 11, 12, 19, 20], [4, 5, 6, 7, 8, 9], [13, 14, 15, 16, 17, 18]]
    assert p.handoff_stage == 1
    assert p.write_slots(p.stages[1]) == [2, 3, 10, 11, 12, 13, 14]
    assert p.write_slots(p.stages[3]) == [-1] * 6
    assert p.renoised_frames(2) == [4, 9] and p.renoised_frames(3) == [13, 18] and p.renoised_frames(1) == []
    assert p.stages == stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS)


def test_i2v_stage_tables():
    p = StagePlan("i2v")
    assert p.stages == [[0], [1], [2, 3, 10, 11, 12, 19, 20], [4, 5, 6, 7, 8, 9], [13, 14, 15, 16, 17, 18]]
    assert p.handoff_stage == 2 and not p.hides_anchors(3) and p.renoised_frames(3) == []


def test_slot_map():
    assert [slot_of(f) for f in (0, 12, 19, 20)] == [0, 12, 13, 14]
    assert max(slot_of(f) for f in range(21) if f not in range(13, 19)) == 14


def test_visibility_evolution_matches_oracle_and_appendix_a():
    p = StagePlan("t2v")
    v, o = VisibleFrames(), stage_ref.VisIndex()
    lkv = []
    for si, frames in enumerate(p.stages):
        if p.hides_anchors(si):
            v.hide(); o.hide()
        if p.shows_anchors(si):
            v.show(); o.show()
        v.on_forward(frames); o.on_forward(frames)
        assert v.slots() == o.slots()
        own = 0 if p.write_slots(frames)[0] >= 0 else len(frames)
        lkv.append(len(v.slots()) + own)
    assert lkv == [2, 9, 13, 21]
    assert [(len(f), k) for f, k in zip(p.stages, lkv)] == T2V_STAGE_SHAPES
    assert v.token_offsets(1560)[-2:] == [31200, 29640] or sorted(v.token_offsets(1560))[-2:] == [29640, 31200]


@pytest.mark.parametrize("model,S,expect_tf", [("14B", 3600, [217.62, 1281.90, 1353.58, 1863.18]),
                                               ("14B", 1560, [83.87, 391.25, 383.20, 478.89]),
                                               ("1.3B", 1560, [9.89, 56.60, 59.28, 80.81])])
def test_flop_table_known_answers(model, S, expect_tf):
    got = [dit_forward_flops(WAN_CONFIGS[model], S, q, kv) / 1e12 for q, kv in T2V_STAGE_SHAPES]
    for g, e in zip(got, expect_tf):
        assert abs(g - e) / e < 2e-3, (got, expect_tf)


def test_bench_chunk_time_assembly():
    from mmpl_amd.stage_plan import T2V_STAGE_SHAPES, assemble_chunk_seconds, dit_forward_flops
    from mmpl_amd.synthetic import WAN_CONFIGS
    fl = [dit_forward_flops(WAN_CONFIGS["14B"], 3600, q, kv) for q, kv in T2V_STAGE_SHAPES]
    true = [0.5, 2.4, 2.6, 3.6]
    # K = 8 from warm-up 4: every stage twice
    st, chunk = assemble_chunk_seconds([true[(4 + i) % 4] for i in range(8)], 4, fl)
    assert st == true and abs(chunk - 51 * sum(true)) < 1e-9
    # K = 5 from warm-up 1: stage 1 twice, no bias from the uneven mix
    st, chunk = assemble_chunk_seconds([true[(1 + i) % 4] for i in range(5)], 1, fl)
    assert st == true
    # K = 2: stages 0 and 1 measured, 2 and 3 priced at the measured FLOP rate (exact when time is proportional to FLOPs)
    prop = [f / 1e15 for f in fl]
    st, chunk = assemble_chunk_seconds(prop[:2], 0, fl)
    assert all(abs(a - b) < 1e-9 for a, b in zip(st, prop))


def test_concurrent_cfg_rule():
    """mmpl_amd.stage_plan.concurrent_cfg_pays: parallel CFG branches for the stages where it was measured to pay (query rows x dim <=
    60 M: all of Wan 1.3B at 480p, 14B at 480p, the 2-frame stage of 14B at 720p), sequential for the long stages of 14B at 720p."""
    from mmpl_amd.stage_plan import concurrent_cfg_pays as pays
    assert all(pays(n * 1560, 1536) for n in (1, 2, 6, 7))                  # Wan 1.3B / 480p
    assert all(pays(n * 1560, 5120) for n in (2, 6, 7))                     # Wan 14B / 480p
    assert pays(2 * 3600, 5120) and not pays(6 * 3600, 5120) and not pays(7 * 3600, 5120)   # Wan 14B / 720p


def test_cross_kv_is_a_pair_with_a_row_count():
    import torch
    from mmpl_amd.dit import CrossKV
    k, v = torch.zeros(2, 3), torch.ones(2, 3)
    kv = CrossKV(k, v, 48)
    a, b = kv
    assert a is k and b is v and kv.rows == 48 and len(kv) == 2 and kv[0] is k


def test_executed_flops_differ_from_algorithmic_only_where_launches_are_skipped():
    """bench.py prints `executed_pflops_per_gpu` beside the algorithmic `achieved_pflops_per_gpu`: the same count when the
    text cross-attention runs over all 512 keys and nothing is shared; minus the collapsed keys' QK^T + PV, minus block 0's
    self-attention + o-projection for the forward that takes them from its CFG partner (mmpl_dit_forward share_in)."""
    from mmpl_amd.stage_plan import T2V_STAGE_SHAPES, dit_forward_flops, dit_forward_flops_executed
    from mmpl_amd.synthetic import WAN_CONFIGS
    cfg, S = WAN_CONFIGS["14B"], 3600
    d, L = cfg["dim"], cfg["num_layers"]
    for q, kv in T2V_STAGE_SHAPES:
        alg = dit_forward_flops(cfg, S, q, kv)
        assert dit_forward_flops_executed(cfg, S, q, kv, 512) == alg
        Lq, Lkv = q * S, kv * S
        assert abs(dit_forward_flops_executed(cfg, S, q, kv, 65) - (alg - L * 4.0 * Lq * (512 - 65) * d)) < 1e-6 * alg
        shared = dit_forward_flops_executed(cfg, S, q, kv, 512, block0_self_attn_shared=True)
        assert abs(shared - (alg - (4.0 * Lq * Lkv * d + 2.0 * Lq * d * d))) < 1e-6 * alg
    # the 14B / 720p step: both savings together are ~1.4 % of the algorithmic count
    alg = sum(2 * dit_forward_flops(cfg, S, q, kv) for q, kv in T2V_STAGE_SHAPES)
    exe = sum(dit_forward_flops_executed(cfg, S, q, kv, 65) + dit_forward_flops_executed(cfg, S, q, kv, 65, block0_self_attn_shared=(q * S * d > 60e6))
              for q, kv in T2V_STAGE_SHAPES)
    assert 0.010 < 1 - exe / alg < 0.018, 1 - exe / alg


def test_attention_history_state_machine():
    """oracle/attn_history_ref.py restates the state byte of the self-attention kernel's pass history (include/mmpl_hip.h `attn_history`);
    tests/test_attn_history_gpu.py holds the kernel's bytes to it launch by launch.  Here: its invariants over all 256 byte values."""
    from oracle.attn_history_ref import MEM, next_state, passes_paid, plan
    assert plan(0) == (True, False) and next_state(0, False) == 0              # a block that never fails never leaves 0
    assert next_state(0, True) == MEM and plan(MEM) == (True, True)            # first failure: references, FAST again (on them)
    for s in range(256):
        try_fast, mem = plan(s)
        for failed in (False, True):
            n = next_state(s, failed)
            assert 0 <= n < 256 and (n & MEM) >= (s & MEM) and ((n & MEM) or not (failed and try_fast))   # bit 7 is never cleared; a failed FAST pass sets it
            assert ((n >> 5) & 3) <= 2 or not try_fast                                           # the level saturates at 2
            if not try_fast:
                assert n == s - 1 and (n & 31) == (s & 31) - 1 and passes_paid(s, failed) == 1.66
            elif not failed:
                assert n == (s & MEM) and passes_paid(s, failed) == 1.0
    # a block that keeps failing even on remembered references: 8 launches to the first retry, 16 to the next, then 31
    s, trace = 0, []
    for launch in range(80):
        tf, _ = plan(s)
        trace.append("F" if tf else "g")
        s = next_state(s, fast_failed=True)
    assert "".join(trace) == "FF" + "g" * 7 + "F" + "g" * 15 + "F" + ("g" * 30 + "F") * 1 + "g" * 23, "".join(trace)
    # ... and a heavy-tailed block whose remembered references hold: pays twice once, then one pass per launch, for ever
    s, cost = 0, []
    for launch, failed in enumerate([True] + [False] * 9):
        cost.append(passes_paid(s, failed))
        s = next_state(s, failed)
    assert cost == [2.66] + [1.0] * 9 and s == MEM
