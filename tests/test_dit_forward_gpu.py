"""GPU parity of the whole HIP DiT forward (C ABI mmpl_dit_forward) against the oracle and the reference's golden
outputs, stage by stage with a live KV cache (-m gpu).

Stated bf16 tolerance (SURVEY.md 8c "tolerance guidance"): per forward, rel-L2(HIP, reference-bf16) <= 2e-2; the
measured value on these cases is printed (typically ~3e-3, i.e. accumulation-order noise of bf16 modules).
"""
import pytest
import torch

from tests.util import GOLDEN, max_abs, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
TOL = 2e-2


def _run_stages(cfg_name, lat, weight_seed, ctx_seed, noise_seed, n_valid, tvals, vis_orders=None):
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    from oracle import stage_ref
    from oracle import wan_dit_ref as W
    cfg = WAN_CONFIGS[cfg_name]
    sd = dit_state_dict(cfg, seed=weight_seed)
    eng = DitEngine(cfg, lat[0], lat[1], "cuda:0")
    eng.load_state_dict(sd)
    ocfg = W.DitCfg(**cfg)
    S = eng.S
    ctx = philox_normal([512, cfg["text_dim"]], ctx_seed)
    ctx[n_valid:] = 0
    noise = philox_normal([1, 21, 16, lat[0], lat[1]], noise_seed)
    kc, vc = eng.new_kv_cache(15)
    ck, cv = eng.precompute_context(ctx.cuda())
    okv = W.new_kv_cache(ocfg, 15, S)
    ocross = [None] * cfg["num_layers"]
    vis = stage_ref.VisIndex()
    outs = []
    for si, frames in enumerate(stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS)):
        if si == 2:
            vis.hide()
        if si == 3:
            vis.show()
        vis.on_forward(frames)
        order = vis_orders[si] if vis_orders else vis.slots()
        x = noise[0, frames].contiguous()
        t = torch.full([len(frames)], tvals[si], dtype=torch.float32)
        y = eng.forward(x.cuda(), t.cuda(), frames, stage_ref.write_slots_for(frames), order, kc, vc, ck, cv)
        torch.cuda.synchronize()
        yo = W.dit_forward(sd, ocfg, x.permute(1, 0, 2, 3), t.view(1, -1), ctx, okv, ocross, frames,
                           stage_ref.write_slots_for(frames), order).permute(1, 0, 2, 3)
        outs.append((y.cpu(), yo))
    return outs, (kc, vc, okv), S


def test_forward_vs_oracle_small_geometry():
    outs, (kc, vc, okv), S = _run_stages("tiny", (16, 24), 3, 4, 5, 20, [999.0, 640.0, 250.0, 0.0])
    for si, (y, yo) in enumerate(outs):
        e = rel_l2(y, yo)
        print(f"stage {si}: rel_l2(HIP, oracle) = {e:.3e} max|d| = {max_abs(y, yo):.3e}")
        assert torch.isfinite(y.float()).all()
        assert e < TOL
    # KV cache contents after the four stages (layer 1, every slot)
    for name, dev_c in (("k", kc), ("v", vc)):
        e = rel_l2(dev_c[1].view(15 * S, -1), okv[1][name].view(15 * S, -1))
        assert e < TOL, (name, e)


def test_forward_vs_reference_golden_480p():
    fx = torch.load(f"{GOLDEN}/dit_forward_tiny.pt")
    m = fx["meta"]
    orders = [fx[f"s{i}_vis_order"] for i in range(4)]
    outs, (kc, vc, okv), S = _run_stages(m["cfg"], (60, 104), m["weight_seed"], m["ctx_seed"], m["noise_seed"], m["n_valid"],
                                         m["tvals"], orders)
    assert S == 1560
    for si, (y, yo) in enumerate(outs):
        ref = fx[f"s{si}_strided"][0]
        e = rel_l2(y[..., ::2, ::2], ref)
        print(f"stage {si}: rel_l2(HIP, reference golden) = {e:.3e}; oracle-vs-golden = {rel_l2(yo[..., ::2, ::2], ref):.3e}")
        assert e < TOL
    e = rel_l2(kc[1, 13 * S:14 * S:13].reshape(-1), fx["kv_l1_slot13_k"].reshape(-1))
    assert e < TOL, e


def test_small_config_three_layers_four_heads():
    outs, _, _ = _run_stages("small", (12, 20), 8, 9, 10, 33, [980.0, 500.0, 100.0, 0.0])
    for si, (y, yo) in enumerate(outs):
        e = rel_l2(y, yo)
        print(f"stage {si}: rel_l2 = {e:.3e}")
        assert e < TOL


def test_hipgraph_replay_matches_eager():
    """One denoise-loop forward captured as a hipGraph replays bit-identically, including after the inputs and the
    timestep were updated in place (the KV cache is rewritten by every replay exactly like an eager call)."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    eng = DitEngine(cfg, 16, 24, "cuda:0")
    eng.load_state_dict(dit_state_dict(cfg, seed=3))
    ctx = philox_normal([512, cfg["text_dim"]], 4)
    ck, cv = eng.precompute_context(ctx.cuda())
    frames, ws, vis = [2, 3, 10, 11, 12, 19, 20], [2, 3, 10, 11, 12, 13, 14], [0, 1, 2, 3, 10, 11, 12, 13, 14]
    outs = {}
    for mode in ("eager", "graph"):
        kc, vc = eng.new_kv_cache(15)
        kc.copy_(philox_normal(list(kc.shape), 9).cuda())
        vc.copy_(philox_normal(list(vc.shape), 10).cuda())
        x = philox_normal([7, 16, 16, 24], 11).cuda()
        t = torch.full([7], 900.0, device="cuda")
        out = torch.empty_like(x)
        g = eng.capture(x, t, frames, ws, vis, kc, vc, ck, cv, out) if mode == "graph" else None
        res = []
        for step, tv in enumerate((900.0, 500.0, 0.0)):
            t.fill_(tv)
            x.copy_(philox_normal([7, 16, 16, 24], 20 + step).cuda())
            if g is not None:
                g.replay()
            else:
                eng.forward(x, t, frames, ws, vis, kc, vc, ck, cv, out=out)
            res.append(out.clone())
        torch.cuda.synchronize()
        outs[mode] = (res, kc.clone())
    for a, b in zip(outs["eager"][0], outs["graph"][0]):
        assert torch.equal(a, b)
    assert torch.equal(outs["eager"][1], outs["graph"][1])


def test_forward_runtime_errors_raise():
    """Error behaviour at the forward seam: bad slot tables / undersized workspace -> RuntimeError with the C-side message,
    and the engine stays usable afterwards."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    eng = DitEngine(cfg, 16, 24, "cuda:0")
    x = philox_normal([2, 16, 16, 24], 1).cuda()
    t = torch.full([2], 500.0, device="cuda")
    kc, vc = eng.new_kv_cache(15)
    with pytest.raises(RuntimeError, match="weights not bound"):
        eng.forward(x, t, [0, 1], [0, 1], [0, 1], kc, vc, kc[:, :512], vc[:, :512])
    eng.load_state_dict(dit_state_dict(cfg, seed=2))
    ck, cv = eng.precompute_context(philox_normal([512, cfg["text_dim"]], 3).cuda())
    with pytest.raises(RuntimeError, match="visible slot out of range"):
        eng.forward(x, t, [0, 1], [0, 1], [0, 99], kc, vc, ck, cv)
    with pytest.raises(RuntimeError, match="write slot out of range"):
        eng.forward(x, t, [0, 1], [0, 15], [0, 1], kc, vc, ck, cv)
    with pytest.raises(RuntimeError, match="all >= 0 or all -1"):
        eng.forward(x, t, [0, 1], [0, -1], [0, 1], kc, vc, ck, cv)
    with pytest.raises(RuntimeError, match="too many"):
        eng.forward(x, t, [0, 1], [0, 1], list(range(15)) * 2, kc, vc, ck, cv)
    ws = eng.workspace(2)
    eng._ws[2] = ws[:1024]
    with pytest.raises(RuntimeError, match="workspace too small"):
        eng.forward(x, t, [0, 1], [0, 1], [0, 1], kc, vc, ck, cv)
    eng._ws[2] = ws
    y = eng.forward(x, t, [0, 1], [0, 1], [0, 1], kc, vc, ck, cv)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()


def test_build_then_smoke_in_one_process():
    """__graft_entry__.build() followed by smoke() in ONE fresh process (the library is dlopen'ed before anything touched
    the device): regression test for the HIP-runtime load order (mmpl_amd/_lib.py imports torch before dlopen)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "[smoke]" in r.stdout


@pytest.mark.parametrize("n_valid", [0, 1, 37, 63, 64, 65, 448, 510, 511, 512])
def test_cross_attention_padded_tail_collapse(n_valid):
    """The zero-padded tail of the text context is ONE repeated K / V row after the text embedding (utils/wan_wrapper.py:46-47 zeroes
    it, causal_fps_model.py:780 / model.py:189 attend over it unmasked): a forward that is TOLD so (`cross_rows` = the count
    mmpl_dit_precompute_context reported for these contents) attends over n_valid + 1 keys with the last one weighted 512 - n_valid
    times (api.hip).  The count is explicit data (it used to be remembered per cross_k POINTER): the same forward without
    it attends over all 512 keys -- equal up to the bf16 rounding of the padded key's P entry (bf16(c * p) vs c * bf16(p): up to
    2 * 2^-9 of that key's share of the output, which is nearly all of it when n_valid is 0 or 1); a COPY of the K / V with the
    count handed on gives the collapsed bits; n_valid >= 511 leaves nothing to collapse."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    eng = DitEngine(cfg, 16, 24, "cuda:0")
    eng.load_state_dict(dit_state_dict(cfg, seed=3))
    ctx = philox_normal([512, cfg["text_dim"]], 40 + n_valid)
    ctx[n_valid:] = 0
    kv = eng.precompute_context(ctx.cuda())
    ck, cv = kv
    assert kv.rows == (n_valid if n_valid < 511 else 512), (kv.rows, n_valid)
    ck2, cv2 = ck.clone(), cv.clone()
    # the padded K / V rows really are one row (what the collapse relies on), in every layer
    if n_valid < 511:
        assert (ck[:, n_valid:] == ck[:, 511:]).all() and (cv[:, n_valid:] == cv[:, 511:]).all()
    frames = [0, 1]
    x = philox_normal([2, 16, 16, 24], 7).cuda()
    t = torch.full([2], 700.0, dtype=torch.float32).cuda()
    outs = []
    for k_, v_, rows in ((ck, cv, kv.rows), (ck2, cv2, None), (ck2, cv2, kv.rows), (ck, cv, 512)):
        kc, vc = eng.new_kv_cache(15)
        outs.append(eng.forward(x, t, frames, [0, 1], [0, 1], kc, vc, k_, v_, cross_rows=rows).clone())
    torch.cuda.synchronize()
    e = rel_l2(outs[0], outs[1])
    print(f"n_valid {n_valid}: rel_l2(collapsed, all 512 keys) = {e:.3e}")
    assert torch.isfinite(outs[0].float()).all() and e < 4e-3
    assert torch.equal(outs[0], outs[2])           # the count travels with the contents, not with the address
    assert torch.equal(outs[1], outs[3])           # no count / text_len: all 512 keys, whichever buffer
    if n_valid >= 511:
        assert torch.equal(outs[0], outs[1])


def test_cross_attn_cache_takes_the_collapsed_path_like_the_bench():
    """`CrossAttnCache.fill` used to register a temporary and copy it, so the pipeline never got the
    collapse the bench measured.  Now precompute writes straight into the cache's buffers and the cache carries `rows`."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    from mmpl_amd.wan_wrapper import CrossAttnCache
    cfg = WAN_CONFIGS["tiny"]
    eng = DitEngine(cfg, 16, 24, "cuda:0")
    eng.load_state_dict(dit_state_dict(cfg, seed=3))
    ctx = philox_normal([512, cfg["text_dim"]], 77)
    ctx[48:] = 0
    cache = CrossAttnCache(eng)
    assert cache.rows == 512 and not cache.is_init
    cache.fill(ctx.cuda())
    kv = eng.precompute_context(ctx.cuda())
    assert cache.is_init and cache.rows == 48 == kv.rows
    assert torch.equal(cache.k_all, kv[0]) and torch.equal(cache.v_all, kv[1])
    cache.fill(philox_normal([512, cfg["text_dim"]], 78).cuda())        # a full-length prompt into the SAME buffers: the stale count must go
    assert cache.rows == 512


def test_forward_bits_do_not_depend_on_where_the_workspace_lies():
    """A forward whose self-attention reads KV-cache slots AND its own scratch pages (write_slots = -1: the in-fill stage that appends
    the stage's K/V to the page table, causal_fps_model.py:254-264) walks its KV tiles in an order that must not depend on whether the
    caching allocator put the workspace below or above the KV cache: same math either way, but another tile order is another fp32
    summation order.  (That is how the two-rank wavefront test -- fresh child processes vs the long-lived pytest process -- once
    lost its bit-identity.)  The launcher orders pages by (allocation group, address): cache slots first, then the scratch pages.
    Both layouts are carved from ONE buffer here, so the addresses are under the test's control."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    lat = (16, 24)
    eng = DitEngine(cfg, lat[0], lat[1], "cuda:0")
    eng.load_state_dict(dit_state_dict(cfg, seed=5))
    S, L, d = eng.S, eng.L, eng.dim
    ctx = philox_normal([512, cfg["text_dim"]], 31)
    ctx[40:] = 0
    ck, cv = eng.precompute_context(ctx.cuda())
    n_slots, nF = 15, 6
    kv_bytes = L * n_slots * S * d * 2
    ws_bytes = eng.workspace(nF).numel()
    pad = lambda n: (n + 4095) // 4096 * 4096
    arena = torch.zeros(2 * pad(kv_bytes) + pad(ws_bytes) + 4096, dtype=torch.uint8, device="cuda:0")
    fill_k = philox_normal([L, n_slots * S, d], 41).to(BF).cuda()
    fill_v = philox_normal([L, n_slots * S, d], 42).to(BF).cuda()
    x = philox_normal([nF, 16, lat[0], lat[1]], 43).to(BF).cuda()
    t = torch.full([nF], 500.0, dtype=torch.float32, device="cuda:0")
    frames, visible = [11, 12, 13, 14, 15, 16], [0, 1, 2, 3, 4, 5, 6, 8, 9]
    outs = []
    for ws_first in (False, True):
        off = 0
        parts = {}
        for name, nb in ((("ws", ws_bytes), ("k", kv_bytes), ("v", kv_bytes)) if ws_first else (("k", kv_bytes), ("v", kv_bytes), ("ws", ws_bytes))):
            parts[name] = arena[off:off + nb]
            off += pad(nb)
        kc = parts["k"].view(BF).view(L, n_slots * S, d)
        vc = parts["v"].view(BF).view(L, n_slots * S, d)
        kc.copy_(fill_k)
        vc.copy_(fill_v)
        assert (parts["ws"].data_ptr() < kc.data_ptr()) == ws_first
        y = eng.forward(x, t, frames, [-1] * nF, visible, kc, vc, ck, cv, workspace=parts["ws"])
        torch.cuda.synchronize()
        outs.append(y.clone())
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0], outs[1])


def test_attention_redo_counters_blocks_and_waves():
    """mmpl_dit_set_attn_stats: {query blocks run, blocks whose max-free FAST pass failed and were redone, WAVES (64 of a block's 256
    rows) that held a failing row themselves}.  Default weights: nothing fails.  QK-norm gains x 12 (logit spread far beyond the FAST
    window): blocks fail, each failing block has 1..4 failing waves, and the forward is still finite (the GENERAL pass took over)."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    frames = [0, 1]
    x = philox_normal([2, 16, 16, 24], 7).cuda()
    t = torch.full([2], 700.0, dtype=torch.float32).cuda()
    ctx = philox_normal([512, cfg["text_dim"]], 41)
    seen = {}
    for gain in (1.0, 12.0):
        sd = dit_state_dict(cfg, seed=3)
        for l in range(cfg["num_layers"]):
            for k in ("self_attn.norm_q.weight", "self_attn.norm_k.weight"):
                sd[f"blocks.{l}.{k}"] = (sd[f"blocks.{l}.{k}"].float() * gain).to(torch.bfloat16)
        eng = DitEngine(cfg, 16, 24, "cuda:0")
        eng.load_state_dict(sd)
        with pytest.raises(RuntimeError, match="never called"):
            eng.read_attn_stats()
        eng.enable_attn_stats()
        kv = eng.precompute_context(ctx.cuda())
        kc, vc = eng.new_kv_cache(15)
        y = eng.forward(x, t, frames, [0, 1], [0, 1], kc, vc, kv[0], kv[1], cross_rows=kv.rows)
        torch.cuda.synchronize()
        blocks, redone, waves, predicted, remembered = eng.read_attn_stats(reset=True)
        assert eng.read_attn_stats() == (0, 0, 0, 0, 0) and predicted == 0 and remembered == 0    # (no history was handed in)
        assert torch.isfinite(y.float()).all()
        n_qb = -(-2 * eng.S // 256)
        items = cfg["num_layers"] * cfg["num_heads"] * n_qb                # (a split-KV tail round runs a query block as 2-4 blocks)
        assert items <= blocks <= 4 * items, (blocks, items)
        assert 0 <= redone <= blocks and redone <= waves <= 4 * redone
        seen[gain] = (blocks, redone, waves)
        eng.disable_attn_stats()
        eng.forward(x, t, frames, [0, 1], [0, 1], kc, vc, kv[0], kv[1], cross_rows=kv.rows)
        torch.cuda.synchronize()
        assert eng.read_attn_stats() == (0, 0, 0, 0, 0)               # off: later eager forwards do not count
    print("attention redo counters {gain: (blocks, blocks redone, waves with a failing row)}:", seen)
    assert seen[1.0][1] == 0 and seen[12.0][1] > 0


@pytest.mark.parametrize("persist", [True, False])
def test_shared_block0_self_attention_is_bit_identical(persist):
    """mmpl_dit_forward share_out / share_in: block 0's self-attention sees nothing that differs between the two CFG branches (same
    latents, same timestep, the same layer-0 K / V in both caches; the text context enters after it).  The cond forward leaves x
    after that residual in `share_out`; the uncond forward given it as `share_in` skips block 0's attention and output projection.
    Against the uncond forward computed in full: same flow, same cache contents (its own layer-0 K / V slots are still written),
    bit for bit -- for a stage that persists its K / V and for the one that attends its own scratch pages."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    cfg = WAN_CONFIGS["tiny"]
    eng = DitEngine(cfg, 16, 24, "cuda:0")
    eng.load_state_dict(dit_state_dict(cfg, seed=3))
    frames = [4, 5, 6]
    ws_slots = [4, 5, 6] if persist else [-1, -1, -1]
    vis = [0, 1, 4, 5, 6] if persist else [0, 1, 2]
    x = philox_normal([3, 16, 16, 24], 7).cuda()
    t = torch.full([3], 433.0, dtype=torch.float32).cuda()
    kv_c = eng.precompute_context(philox_normal([30, cfg["text_dim"]], 1).cuda())
    kv_u = eng.precompute_context(philox_normal([9, cfg["text_dim"]], 2).cuda())

    def caches(seed):
        kc, vc = eng.new_kv_cache(15)
        kc.copy_(philox_normal(list(kc.shape), seed).cuda())
        vc.copy_(philox_normal(list(vc.shape), seed + 1).cuda())
        return kc, vc
    kc_c, vc_c = caches(10)
    outs = {}
    for mode in ("full", "shared"):
        kc_u, vc_u = caches(20)
        kc_u[0].copy_(kc_c[0])                      # the precondition: layer 0 of the two caches agrees (deeper layers and contexts do not)
        vc_u[0].copy_(vc_c[0])
        kc1, vc1 = kc_c.clone(), vc_c.clone()
        share = eng.shared_block0_buffer(3) if mode == "shared" else None
        yc = eng.forward(x, t, frames, ws_slots, vis, kc1, vc1, kv_c[0], kv_c[1], cross_rows=kv_c.rows, share_out=share).clone()
        yu = eng.forward(x, t, frames, ws_slots, vis, kc_u, vc_u, kv_u[0], kv_u[1], cross_rows=kv_u.rows, share_in=share).clone()
        torch.cuda.synchronize()
        outs[mode] = (yc, yu, kc_u.clone(), vc_u.clone())
    assert torch.equal(outs["full"][0], outs["shared"][0])                  # the producer's own result is untouched
    assert torch.equal(outs["full"][1], outs["shared"][1]), rel_l2(outs["shared"][1], outs["full"][1])
    assert torch.equal(outs["full"][2], outs["shared"][2]) and torch.equal(outs["full"][3], outs["shared"][3])
    assert not torch.equal(outs["full"][0], outs["full"][1])                # (the branches do differ: another context)
    with pytest.raises(RuntimeError, match="exclusive"):
        eng.forward(x, t, frames, ws_slots, vis, kc_c, vc_c, kv_c[0], kv_c[1], share_out=eng.shared_block0_buffer(3), share_in=eng.shared_block0_buffer(3))
