"""The self-attention kernel's pass history (`attn_history`, include/mmpl_hip.h) and the MMPL_CHECK_SHARE guard (-m gpu).

attn_w64_kernel runs a max-free FAST softmax pass per 256-row query block and redoes a block whose row sums left the window with
the GENERAL pass.  With a history (a state byte + 128 lane references per block) a block that failed takes its next FAST reference from
what the last pass learned; a block that fails even so goes straight to GENERAL and FAST is retried on the 8th launch after that
failure (then the 16th, then the 31st).  Checker: `oracle.sdpa_fp32` for every launch -- the
result is the exact softmax whichever pass ran -- plus the counters the kernel keeps.  Replaces attention.py:139-185 + the gather of
causal_fps_model.py:219-227, as every attention test.
"""
import ctypes as C
import math
import os
import subprocess
import sys

import pytest
import torch

from tests.util import max_abs, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sp():
    from mmpl_amd import _lib
    return _lib.stream_ptr()


def _spiked_case(split_tail: bool, heavy_heads=()):
    """q / K / V with a few keys that are large multiples of a few queries (scores no FAST pass can hold) in known query blocks;
    heavy_heads: instead, those heads' K rows scaled x 4 in the first page and x 45 in the later ones: every query row's scores have a
    standard deviation of ~6 (log2 units) over the keys the FAST pass samples its reference from and ~65 over the rest, so the row
    log-sum-exps (~210 +- 27) lie far above anything the sample predicts -- every block of the head fails the sampled reference, like
    under a large QK-norm gain -- while two rows of a lane stay well within the window of each other (their spread is what a shared
    reference cannot absorb: at +-41, a scale of 70, one block in ten fails on the remembered reference too)."""
    torch.manual_seed(12)
    dev = "cuda:0"
    H, S, n_pages = 8, 640, 3
    Lq = 256 * 41 - 57 if split_tail else 256 * 6 + 40          # 41 blocks x 8 heads: a split-KV tail round; 7 blocks: none
    d = H * 128
    q32 = torch.randn(Lq, d, device=dev)
    kc32 = torch.randn(n_pages * S, d, device=dev)
    vc = torch.randn(n_pages * S, d, device=dev).to(BF)
    hot_rows = [5, 300, Lq - 700, Lq - 3]
    hot_keys = [17, 700, 1300, 1900]
    hot_heads = (0, 3, 7)
    if heavy_heads:
        for h in heavy_heads:
            kc32[:S, h * 128:(h + 1) * 128] *= 4.0
            kc32[S:, h * 128:(h + 1) * 128] *= 45.0
        hot_heads = tuple(heavy_heads)
    else:
        for r, kk in zip(hot_rows, hot_keys):
            for h in hot_heads:
                kc32[kk, h * 128:(h + 1) * 128] = q32[r, h * 128:(h + 1) * 128] * 25.0
    c = (1.0 / math.sqrt(128)) * 1.4426950408889634
    q = (q32 * c).to(BF)
    return dict(H=H, S=S, n_pages=n_pages, Lq=Lq, d=d, q=q, q_ref=q.float() / c, kc=kc32.to(BF), kc_calm=torch.randn(n_pages * S, d, device=dev).to(BF), vc=vc,
                hot_heads=set(hot_heads))


def _launch(lib, cs, kc, hist, stats, ws):
    from mmpl_amd import _lib
    n_pages, S, d = cs["n_pages"], cs["S"], cs["d"]
    kp = (C.c_void_p * n_pages)(*[kc[i * S:].data_ptr() for i in range(n_pages)])
    vp = (C.c_void_p * n_pages)(*[cs["vc"][i * S:].data_ptr() for i in range(n_pages)])
    o = torch.full((cs["Lq"], d), float("nan"), device="cuda:0", dtype=BF)
    stats.zero_()
    _lib.check(lib.mmpl_attn_fwd_history(_lib.ptr(cs["q"]), d, _lib.ptr(o), d, kp, vp, d, d, n_pages, S, cs["Lq"], cs["H"], 1.0 / math.sqrt(128),
                                         _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.ptr(hist), _lib.ptr(stats), _sp()))
    torch.cuda.synchronize()
    return o, [int(v) for v in stats.cpu()]


def _ref(cs, kc, rows):
    from oracle import wan_dit_ref as W
    H, d = cs["H"], cs["d"]
    return W.sdpa_fp32(cs["q_ref"][rows].reshape(1, -1, H, 128).cpu(), kc.float().reshape(1, -1, H, 128).cpu(),
                       cs["vc"].float().reshape(1, -1, H, 128).cpu()).reshape(-1, d)


def _states(cs, hist):
    """the state bytes [H, n_qb, 4 parts] (the lane references follow them, include/mmpl_hip.h)"""
    n_qb = (cs["Lq"] + 255) // 256
    return hist[:cs["H"] * n_qb * 4].cpu().view(cs["H"], n_qb, 4)


MEM = 128      # state byte, bit 7: the lane references are valid


@pytest.mark.parametrize("split_tail", [False, True])
def test_history_backs_off_blocks_that_fail_even_on_remembered_references(lib, split_tail):
    """Single spiked keys: one query row of a lane sees a score ~400 (log2 units) above everything its partner row sees, so no shared
    reference holds both -- the FAST pass fails on the sampled reference AND on the remembered one.  The state byte then sends the
    block straight to the GENERAL pass (nothing paid twice) and retries FAST on the 8th launch after the failure, then the 16th; when
    the retry holds (calm keys) the back-off clears.  Every launch is checked against fp32."""
    cs = _spiked_case(split_tail)
    dev = "cuda:0"
    n_hist = lib.mmpl_attn_history_bytes(cs["Lq"], cs["H"])
    n_qb = (cs["Lq"] + 255) // 256
    n_state = cs["H"] * n_qb * 4
    assert n_hist == (n_state + 255) // 256 * 256 + n_state * 128 * 2
    hist = torch.zeros(n_hist, dtype=torch.uint8, device=dev)
    stats = torch.zeros(5, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev) if split_tail else None
    rows = torch.cat([torch.arange(0, 600), torch.arange(cs["Lq"] - 1200, cs["Lq"])])
    ref = _ref(cs, cs["kc"], rows)
    nz = lambda: sorted(set(_states(cs, hist)[_states(cs, hist) != 0].tolist()))
    from oracle.attn_history_ref import next_state, plan       # the state byte's CPU restatement: a block with a spiked row fails whenever FAST runs
    model = 0

    # stateless launch = the kernel of every earlier round: the reference bits for "FAST then GENERAL"
    o_stateless, st0 = _launch(lib, cs, cs["kc"], None, stats, ws)
    assert st0[1] > 0 and st0[3] == 0 and st0[4] == 0

    # launch 1 on a zeroed history: the same passes as the stateless launch, the same bits; the failing blocks now hold references
    o1, st = _launch(lib, cs, cs["kc"], hist, stats, ws)
    assert st == st0 and torch.equal(o1, o_stateless)
    h = _states(cs, hist)
    model = next_state(model, True)
    assert int((h != 0).sum()) == st[1] and nz() == [MEM] == [model]
    marked = {(int(a), int(b)) for a, b, _ in (h != 0).nonzero().tolist()}
    # row 300 (query block 1, main round) meets its key in KV tile 10, far outside the FAST reference's four-tile sample
    assert {a for a, _ in marked} == cs["hot_heads"] and all((hh, 1) in marked for hh in cs["hot_heads"]), marked
    redone = st[1]

    # launch 2: FAST on the remembered references; the blocks with a spiked row fail again (most of the marked ones) -> back-off;
    # the others (marked only because of a marginal row) hold
    o, st = _launch(lib, cs, cs["kc"], hist, stats, ws)
    assert st[1] + st[4] == redone and st[1] > 0 and st[3] == 0 and rel_l2(o[rows], ref) < 1e-2, st
    again, held = st[1], st[4]
    model = next_state(model, True)
    assert nz() in ([MEM | 8], [MEM, MEM | 8]) and model == MEM | 8
    # launches 3 .. 9: those blocks go straight to GENERAL (nothing is paid twice), the countdown runs 7 .. 1
    for n in range(3, 10):
        o, st = _launch(lib, cs, cs["kc"], hist, stats, ws)
        assert st[1] == 0 and st[3] == again and st[4] == held and st[0] == st0[0], (n, st)
        assert rel_l2(o[rows], ref) < 1e-2 and torch.isfinite(o.float()).all()
        assert plan(model) == (False, True)
        model = next_state(model, True)
        assert model == (MEM | (10 - n)) and model in nz()
    # launch 10: FAST is tried again, fails again, the interval doubles (level 1, countdown 16)
    o, st = _launch(lib, cs, cs["kc"], hist, stats, ws)
    assert plan(model) == (True, True)
    model = next_state(model, True)
    assert st[1] == again and st[3] == 0 and model == (MEM | (1 << 5) | 16) and model in nz() and rel_l2(o[rows], ref) < 1e-2
    # ... 15 launches straight to GENERAL, and when FAST holds at the retry (calm keys now) the back-off clears
    ref_calm = _ref(cs, cs["kc_calm"], rows)
    for n in range(15):
        o, st = _launch(lib, cs, cs["kc_calm"], hist, stats, ws)
        assert st[1] == 0 and st[3] == again and plan(model) == (False, True)
        model = next_state(model, False)
        assert model in nz()
        # (every launch: the 15th takes the countdown from 2 to 1 -- the value at which a wave that read the byte late would decide
        # differently from its block; the kernel rewrites the byte only after a barrier)
        assert rel_l2(o[rows], ref_calm) < 1e-2, n
    o, st = _launch(lib, cs, cs["kc_calm"], hist, stats, ws)
    assert plan(model) == (True, True) and next_state(model, False) == MEM
    assert st[1] == 0 and st[3] == 0 and st[4] == redone and nz() == [MEM]
    assert rel_l2(o[rows], ref_calm) < 1e-2
    for r in (5, 300):                                                     # the spiked rows on the spiked keys: one-hot-like softmax, 2 bf16 ulps
        i = (rows == r).nonzero()[0, 0]
        assert max_abs(o1[r], ref[i]) < 2.0 ** -7 * ref[i].abs().max().item() + 1e-3


@pytest.mark.parametrize("split_tail", [False, True])
def test_history_remembers_the_reference_of_heavy_tailed_heads(lib, split_tail):
    """Whole heads heavy-tailed (row maxima in the hundreds of log2 units, far above the sampled reference -- what a large QK-norm gain
    does): the sampled FAST reference fails on those heads' blocks, the GENERAL pass redoes them and leaves every lane's log-sum-exp;
    from launch 2 on the FAST pass takes that as its reference and HOLDS -- one pass per block instead of two, no back-off.  Drifting
    scores (q scaled by 0.9 ... 1.1 from launch to launch) are followed by the references.  Checker: fp32, every launch."""
    cs = _spiked_case(split_tail, heavy_heads=(1, 6))
    dev = "cuda:0"
    hist = torch.zeros(lib.mmpl_attn_history_bytes(cs["Lq"], cs["H"]), dtype=torch.uint8, device=dev)
    stats = torch.zeros(5, dtype=torch.int64, device=dev)
    ws = torch.empty(lib.mmpl_attn_workspace_bytes(), dtype=torch.uint8, device=dev) if split_tail else None
    rows = torch.cat([torch.arange(0, 600), torch.arange(cs["Lq"] - 1200, cs["Lq"])])
    q0, qref0 = cs["q"].clone(), cs["q_ref"].clone()
    o1, st1 = _launch(lib, cs, cs["kc"], hist, stats, ws)
    assert st1[1] > 0 and rel_l2(o1[rows], _ref(cs, cs["kc"], rows)) < 1e-2
    h = _states(cs, hist)
    assert {int(a) for a, _, _ in (h != 0).nonzero().tolist()} == cs["hot_heads"]
    n_heavy = st1[1]
    if not split_tail:
        assert n_heavy == len(cs["hot_heads"]) * h.shape[1]                # every block of those heads (split parts that see only late keys hold)
    for step, g in enumerate((1.0, 1.05, 1.1, 1.0, 0.9, 0.95)):
        cs["q"] = (q0.float() * g).to(BF)                                   # the same queries, drifting in scale like consecutive denoise steps
        cs["q_ref"] = cs["q"].float() * (qref0[0, 0] / q0[0, 0].float())
        o, st = _launch(lib, cs, cs["kc"], hist, stats, ws)
        # (nearly) every heavy block: ONE pass, on the remembered references; a lane whose two rows drifted apart may fail and back off
        assert st[1] + st[3] + st[4] == n_heavy and st[4] >= 0.75 * n_heavy, (step, st)
        assert rel_l2(o[rows], _ref(cs, cs["kc"], rows)) < 1e-2, step
        assert MEM in set(_states(cs, hist)[_states(cs, hist) != 0].tolist())


def _heavy_engine(gain=8.0):
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict
    cfg = WAN_CONFIGS["tiny"]
    sd = dit_state_dict(cfg, seed=3)
    for k in list(sd):
        if k.endswith("self_attn.norm_q.weight") or k.endswith("self_attn.norm_k.weight"):
            sd[k] = (sd[k].float() * gain).to(BF)                         # QK-norm gains x8: what makes real checkpoints' heads heavy-tailed
    eng = DitEngine(cfg, 32, 48, "cuda:0")                                # S = 384 tokens per frame
    eng.load_state_dict(sd)
    return eng, cfg


def test_identical_sequences_are_bit_identical_eager_and_graph():
    """The output now depends on the launches before.  Two identical 3-step sequences from a zeroed history -- one eager, one a
    hipGraph replayed three times -- give the same bits step by step (and the history was really in play: blocks were redone on
    step 1 and ran their FAST pass on remembered references on step 2)."""
    from mmpl_amd.synthetic import philox_normal
    eng, cfg = _heavy_engine()
    frames, slots, vis = [2, 3, 4], [2, 3, 4], [0, 1, 2, 3, 4]
    kv = eng.precompute_context(philox_normal([30, cfg["text_dim"]], 1).cuda())
    xs = [philox_normal([3, 16, 32, 48], 40 + i).cuda() for i in range(3)]
    t = torch.full([3], 433.0, dtype=torch.float32).cuda()

    def caches():
        kc, vc = eng.new_kv_cache(15)
        kc.copy_(philox_normal(list(kc.shape), 10).cuda() * 4)
        vc.copy_(philox_normal(list(vc.shape), 11).cuda())
        return kc, vc

    stats = eng.enable_attn_stats()
    per_step = []
    kc, vc = caches()
    hist = eng.new_attn_history(3)
    outs_e = []
    for x in xs:
        stats.zero_()
        outs_e.append(eng.forward(x, t, frames, slots, vis, kc, vc, kv[0], kv[1], cross_rows=kv.rows, attn_history=hist).clone())
        per_step.append(eng.read_attn_stats())
    assert per_step[0][1] > 0 and per_step[0][3] == 0 and per_step[0][4] == 0, per_step     # step 1: FAST failed somewhere, paid twice
    # ... and by step 3 the history decides: blocks run FAST on remembered references or (these wild synthetic scores: a noise cache x 4
    # under gains x 8) fail on them too and go straight to GENERAL
    assert per_step[2][3] + per_step[2][4] > 0 and per_step[2][1] < per_step[0][1], per_step
    kc_e, vc_e = kc.clone(), vc.clone()

    kc, vc = caches()
    hist2 = eng.new_attn_history(3)
    xg = xs[0].clone()
    out = torch.empty_like(outs_e[0])
    g = eng.capture(xg, t, frames, slots, vis, kc, vc, kv[0], kv[1], out, cross_rows=kv.rows, attn_history=hist2)
    hist2.zero_()                                                         # (capture may have run one eager warm-up forward)
    kc2, vc2 = caches()
    kc.copy_(kc2); vc.copy_(vc2)
    for i, x in enumerate(xs):
        xg.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, outs_e[i]), (i, rel_l2(out, outs_e[i]))
    assert torch.equal(kc, kc_e) and torch.equal(vc, vc_e) and torch.equal(hist, hist2)
    # and without a history the forward is what it always was: stateless, every call the same bits
    eng.disable_attn_stats()
    kc, vc = caches()
    a = eng.forward(xs[0], t, frames, slots, vis, kc, vc, kv[0], kv[1], cross_rows=kv.rows).clone()
    b = eng.forward(xs[0], t, frames, slots, vis, kc, vc, kv[0], kv[1], cross_rows=kv.rows).clone()
    assert torch.equal(a, b) and torch.equal(a, outs_e[0])               # step 1 of a zeroed history == the stateless forward


_SHARE_GUARD_CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.environ["MMPL_ROOT"])
from mmpl_amd.dit import DitEngine
from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
cfg = WAN_CONFIGS["tiny"]
eng = DitEngine(cfg, 16, 24, "cuda:0")
eng.load_state_dict(dit_state_dict(cfg, seed=3))
frames, slots, vis = [4, 5, 6], [4, 5, 6], [0, 1, 4, 5, 6]
x = philox_normal([3, 16, 16, 24], 7).cuda()
t = torch.full([3], 433.0, dtype=torch.float32).cuda()
kv_c = eng.precompute_context(philox_normal([30, cfg["text_dim"]], 1).cuda())
kv_u = eng.precompute_context(philox_normal([9, cfg["text_dim"]], 2).cuda())
def caches(seed):
    kc, vc = eng.new_kv_cache(15)
    kc.copy_(philox_normal(list(kc.shape), seed).cuda()); vc.copy_(philox_normal(list(vc.shape), seed + 1).cuda())
    return kc, vc
kc_c, vc_c = caches(10)
kc_u, vc_u = caches(20)
kc_u[0].copy_(kc_c[0]); vc_u[0].copy_(vc_c[0])
share = eng.shared_block0_buffer(3)
def pair():
    eng.forward(x, t, frames, slots, vis, kc_c, vc_c, kv_c[0], kv_c[1], cross_rows=kv_c.rows, share_out=share)
    return eng.forward(x, t, frames, slots, vis, kc_u, vc_u, kv_u[0], kv_u[1], cross_rows=kv_u.rows, share_in=share)
pair(); torch.cuda.synchronize()
assert eng.share_check_failures() == 0
print("OK-equal")
# one element of one visible layer-0 V slot of the uncond cache (a slot this stage does not rewrite)
S = eng.S
vc_u[0, 1 * S + 7, 33] += 1.0
try:
    pair(); torch.cuda.synchronize()
    print("NO-ERROR")
except RuntimeError as e:
    print("RAISED", "MMPL_CHECK_SHARE" in str(e))
assert eng.share_check_failures() == 0          # the failing call reported and reset the count
# inside hipGraph replays nothing can be read back: the mismatch is counted on the device
vc_u[0].copy_(vc_c[0])
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    pair()
g.replay(); torch.cuda.synchronize()
assert eng.share_check_failures() == 0
kc_u[0, 0 * S + 3, 5] += 1.0                    # ... a K slot this time
g.replay(); g.replay(); torch.cuda.synchronize()
print("GRAPH-COUNT", eng.share_check_failures())
'''


def test_share_in_promise_is_checked_under_MMPL_CHECK_SHARE():
    """`share_in` states that layer 0 of the two CFG caches agrees; MMPL_CHECK_SHARE=1 (read once per process: a child) makes the
    forward verify it on the device.  Equal caches pass; a poisoned V element in one branch's visible layer-0 slot makes the eager
    share_in forward raise; inside hipGraph replays the mismatches are counted (`share_check_failures`)."""
    env = dict(os.environ, MMPL_CHECK_SHARE="1", MMPL_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", _SHARE_GUARD_CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK-equal" in r.stdout and "RAISED True" in r.stdout and "GRAPH-COUNT 2" in r.stdout, r.stdout
    # and without the switch nothing is checked (and nothing is allocated or launched for it)
    env.pop("MMPL_CHECK_SHARE")
    r = subprocess.run([sys.executable, "-c", _SHARE_GUARD_CHILD.replace('print("NO-ERROR")', 'print("NO-ERROR"); sys.exit(0)')], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "NO-ERROR" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
