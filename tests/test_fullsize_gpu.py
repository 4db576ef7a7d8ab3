"""Full-size parity (BASELINE.json configs[1] and configs[2]): the HIP DiT forward at the real 1.3B/480p and 14B/720p
shapes, all layers, all four T2V stage patterns with a live KV cache, against the oracle's restatement executed with
the same seeded weights.  At these sizes the CPU oracle needs hours (0.25 TFLOP/s), so the checker here is
the SAME oracle code (oracle/wan_dit_ref.py) evaluated by PyTorch on the device (rocBLAS / SDPA in bf16, heads in
groups so the score matrix stays bounded) -- an implementation that shares nothing with libmmpl_hip.so.  The CPU
oracle itself is pinned to the reference in tests/test_oracle_golden.py; its device evaluation is tied back to the
CPU one on the small case below.  Stated tolerance: rel-L2 <= 1.8e-2 per forward AND <= 1.3 x the value recorded for the
configuration when the bound was last reviewed (RECORDED below) -- a regression of 30 % fails even while it is still
inside the absolute bound."""
import os

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")     # the checker's torch convolutions: skip MIOpen's exhaustive search

import pytest  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from tests.util import rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1.8e-2
# max over the stage patterns of rel_l2(HIP, oracle on device), measured on MI355X (profiles/r03*_gputests_parity.log)
RECORDED = {("1.3B", "t2v"): 1.25e-2, ("14B", "t2v"): 1.5e-2, ("14B", "i2v"): 1.5e-2}


def _bound(key):
    return min(TOL, 1.3 * RECORDED[key])


def _grouped_sdpa(q, k, v, group=4):
    """attention.py:170-185 semantics ([B,L,N,D] bf16 in/out), `group` heads at a time."""
    outs = []
    for h0 in range(0, q.shape[2], group):
        qq, kk, vv = (u[:, :, h0:h0 + group].transpose(1, 2).to(torch.bfloat16) for u in (q, k, v))
        outs.append(F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2))
    return torch.cat(outs, dim=2).contiguous()


def _stages(cfg_name, lat, layers=None, seed=21, mode="t2v", mutate=None, n_stages=None):
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict
    from oracle import stage_ref
    from oracle import wan_dit_ref as W
    dev = "cuda:0"
    cfg = dict(WAN_CONFIGS[cfg_name])
    if layers:
        cfg["num_layers"] = layers
    sd = dit_state_dict(cfg, seed=seed, device=dev)
    if mutate is not None:
        mutate(sd, cfg)
    eng = DitEngine(cfg, lat[0], lat[1], dev)
    eng.load_state_dict(sd)
    ocfg = W.DitCfg(**cfg)
    S = eng.S
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    ctx = torch.randn(512, cfg["text_dim"], generator=g, device=dev).bfloat16()
    ctx[64:] = 0
    noise = torch.randn(21, 16, lat[0], lat[1], generator=g, device=dev).bfloat16()
    kc, vc = eng.new_kv_cache(15)
    ck, cv = eng.precompute_context(ctx)
    okv = [{n: t.to(dev) for n, t in d.items()} for d in W.new_kv_cache(ocfg, 15, S)]
    ocross = [None] * cfg["num_layers"]
    vis = stage_ref.VisIndex()
    errs = []
    clean = stage_ref.T2V_CLEAN_STEPS if mode == "t2v" else stage_ref.I2V_CLEAN_STEPS
    for si, frames in enumerate(stage_ref.stage_frames(clean)[:n_stages]):
        if mode == "t2v" and si == 2:      # I2V never hides frames 19, 20 (MMPL_i2v/pipeline/casual_fps_inference.py:253-335)
            vis.hide()
        if mode == "t2v" and si == 3:
            vis.show()
        vis.on_forward(frames)
        order = vis.slots()
        x = noise[frames].contiguous()
        t = torch.full([len(frames)], (999.0, 750.0, 402.0, 92.0, 0.0)[si], dtype=torch.float32, device=dev)
        ws = stage_ref.write_slots_for(frames)
        y = eng.forward(x, t, frames, ws, order, kc, vc, ck, cv)
        yo = W.dit_forward(sd, ocfg, x.permute(1, 0, 2, 3), t.view(1, -1), ctx, okv, ocross, frames, ws, order,
                           attn_fn=_grouped_sdpa).permute(1, 0, 2, 3)
        torch.cuda.synchronize()
        assert torch.isfinite(y.float()).all()
        errs.append(rel_l2(y, yo))
        # the K/V the stage persisted (RoPE'd keys; values) agree too -- the next stage attends to them
        for slot in [w for w in ws if w >= 0][:1]:
            k_hip = kc[cfg["num_layers"] - 1, slot * S:(slot + 1) * S]
            k_orc = okv[-1]["k"][0, slot * S:(slot + 1) * S].reshape(S, -1)
            assert rel_l2(k_hip, k_orc) < TOL
    return errs


def test_device_evaluated_oracle_equals_cpu_oracle_small():
    """Ties the device evaluation of the oracle to the pinned CPU oracle (same code, two executors)."""
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    from oracle import wan_dit_ref as W
    cfg = WAN_CONFIGS["tiny"]
    ocfg = W.DitCfg(**cfg)
    sd = dit_state_dict(cfg, seed=3)
    x = philox_normal([16, 2, 16, 24], 5)
    ctx = philox_normal([512, cfg["text_dim"]], 6)
    t = torch.full([1, 2], 700.0)
    S = 8 * 12
    a = W.dit_forward(sd, ocfg, x, t, ctx, W.new_kv_cache(ocfg, 15, S), [None] * cfg["num_layers"], [0, 1], [0, 1], [0, 1])
    sdd = {k: v.cuda() for k, v in sd.items()}
    okv = [{n: u.cuda() for n, u in d.items()} for d in W.new_kv_cache(ocfg, 15, S)]
    b = W.dit_forward(sdd, ocfg, x.cuda(), t.cuda(), ctx.cuda(), okv, [None] * cfg["num_layers"], [0, 1], [0, 1], [0, 1],
                      attn_fn=_grouped_sdpa)
    e = rel_l2(b, a)
    print(f"oracle on device vs oracle on CPU (tiny): rel_l2 = {e:.3e}")
    assert e < 5e-3


@pytest.mark.parametrize("cfg_name,lat", [("1.3B", (60, 104)), ("14B", (90, 160))])
def test_full_size_forward_all_stage_patterns(cfg_name, lat):
    errs = _stages(cfg_name, lat)
    print(f"{cfg_name} {lat}: rel_l2(HIP, oracle on device) per stage = " + ", ".join(f"{e:.3e}" for e in errs))
    assert max(errs) < _bound((cfg_name, "t2v"))


def test_full_size_forward_heavy_tailed_14B_anchor_stage():
    """Whole-forward heavy-tail case at 14B / 720p (all 40 layers, stages s0 then s1 = 7 query frames over 9): the statistics real
    checkpoints have and unit-variance synthetic weights do not -- QK-norm gains x8 (sharp softmax rows: the exp2 FAST pass's
    overflow / redo path, bf16 q rounding) and 6 massive-activation channels in the residual stream (patch-embedding bias +-60 on
    6 columns: LayerNorm statistics, GEMM accumulation and the bf16 residual adds all see a 100x dynamic range).  Checker as
    above.  The oracle's own two executors differ more here too (sharp softmax amplifies a 1-ulp score change), hence the
    separately stated bound of 1.2e-2 (measured 6.2e-3 / 5.9e-3: the massive channels dominate both norms)."""
    def mutate(sd, cfg):
        for l in range(cfg["num_layers"]):
            for k in ("self_attn.norm_q.weight", "self_attn.norm_k.weight"):
                sd[f"blocks.{l}.{k}"] = (sd[f"blocks.{l}.{k}"].float() * 8.0).to(torch.bfloat16)
        b = sd["patch_embedding.bias"].float()
        b[[7, 300, 1111, 2049, 3333, 5000]] = torch.tensor([60.0, -60.0, 45.0, -45.0, 60.0, -50.0], device=b.device)
        sd["patch_embedding.bias"] = b.to(torch.bfloat16)
    errs = _stages("14B", (90, 160), mutate=mutate, n_stages=2, seed=33)
    print("14B (90, 160) heavy-tailed (QK gains x8, 6 massive channels): rel_l2(HIP, oracle on device) s0, s1 = " + ", ".join(f"{e:.3e}" for e in errs))
    assert len(errs) == 2 and max(errs) < 1.2e-2


def test_full_size_forward_i2v_stage_patterns():
    """The I2V stage plan at 14B/720p (BASELINE.json configs[4] shapes on one GPU): query / visible frames (1,1) image
    latent, (1,2), (7,9) anchors, (6,15) with 19,20 still visible, (6,21) non-persisting
    (MMPL_i2v/pipeline/casual_fps_inference.py:253-255)."""
    errs = _stages("14B", (90, 160), mode="i2v")
    print("14B (90, 160) i2v: rel_l2(HIP, oracle on device) per stage = " + ", ".join(f"{e:.3e}" for e in errs))
    assert len(errs) == 5 and max(errs) < _bound(("14B", "i2v"))


def test_vae_720p_decode_encode_vs_device_evaluated_oracle():
    """The Wan 3D-VAE at the 720p geometry (latents 90x160 -> 720x1280 px): decode of 2 latent frames (5 px frames) and
    encode of 5 px frames (2 latents), HIP vs oracle/vae_ref.py evaluated by PyTorch on the device (the CPU oracle
    needs ~0.1 PFLOP here).  Tolerance as in tests/test_vae_gpu.py: rel-L2 <= 3e-2."""
    from mmpl_amd.synthetic import vae_state_dict
    from mmpl_amd.vae import VaeEngine
    from oracle import vae_ref
    from tests.test_vae_gpu import MEAN, STD
    dev = "cuda:0"
    sd = vae_state_dict(seed=3)
    eng = VaeEngine(90, 160, dev)
    eng.load_state_dict(sd)
    sdd = {k: v.to(dev) for k, v in sd.items()}
    g = torch.Generator(device=dev).manual_seed(5)
    z = torch.randn(2, 16, 90, 160, generator=g, device=dev).bfloat16()
    out = eng.decode(z, MEAN, STD)                                                  # [5, 3, 720, 1280]
    ref = vae_ref.decode_to_pixel(sdd, z.unsqueeze(0), MEAN, STD)[0]
    torch.cuda.synchronize()
    e_dec = rel_l2(out, ref)
    px = (torch.rand(3, 5, 720, 1280, generator=g, device=dev) * 2 - 1).bfloat16()
    lat = eng.encode(px, MEAN, STD)                                                 # [2, 16, 90, 160]
    ref_lat = vae_ref.encode_to_latent(sdd, px.unsqueeze(0), MEAN, STD)[0]
    torch.cuda.synchronize()
    e_enc = rel_l2(lat, ref_lat)
    print(f"VAE 720p: rel_l2(HIP, oracle on device) decode = {e_dec:.3e}, encode = {e_enc:.3e}")
    assert out.shape == (5, 3, 720, 1280) and lat.shape == (2, 16, 90, 160)
    assert e_dec < 3e-2 and e_enc < 3e-2


def test_t5_production_dims_two_layers_vs_cpu_oracle():
    """umT5-xxl dims (dim 4096, 64 heads, ffn 10240, text_len 512), two layers, against the pinned CPU oracle."""
    from mmpl_amd.synthetic import t5_state_dict
    from mmpl_amd.t5 import T5Engine
    from oracle import t5_ref
    cfg = dict(vocab=2048, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=2, num_buckets=32)
    sd = t5_state_dict(cfg, seed=13)
    L = 512
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(2, cfg["vocab"], (1, L), generator=g)
    mask = torch.zeros(1, L, dtype=torch.long)
    mask[0, :91] = 1
    ids[0, 91:] = 0
    want = t5_ref.text_encoder_forward(sd, ids, mask, cfg["num_heads"], cfg["num_buckets"], cfg["num_layers"])
    eng = T5Engine(cfg, text_len=L, device="cuda:0")
    eng.load_state_dict(sd)
    out = eng.encode(ids, mask)
    torch.cuda.synchronize()
    e = rel_l2(out[0, :91], want[0, :91])
    print(f"umT5 xxl dims, 2 layers: rel_l2(HIP, CPU oracle) = {e:.3e}")
    assert e < 2e-2 and out[0, 91:].abs().sum().item() == 0


def test_i2v_model_type_production_dims():
    """WanModel(model_type='i2v') geometry at real dims: 1.3B width (dim 1536, 12 heads), 480p frames (1560 ragged-tile tokens),
    in_dim 36, 257 CLIP tokens of 1280 through img_emb, 4 layers, 3 frames that write and see each other; checker = the oracle
    code on the device (as above).  The tiny-size forward is pinned to the reference itself in tests/test_i2v_clip_gpu.py."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_i2v_state_dict
    from oracle import wan_dit_ref as W
    dev = "cuda:0"
    cfg = dict(WAN_CONFIGS["1.3B"], num_layers=4, model_type="i2v")
    sd = dit_i2v_state_dict(cfg, seed=31, device=dev)
    eng = DitEngine(cfg, 60, 104, dev, max_frames=3)
    eng.load_state_dict(sd)
    ocfg = W.DitCfg(**{k: v for k, v in dict(cfg, in_dim=36).items() if k != "model_type"})
    g = torch.Generator(device=dev).manual_seed(32)
    ctx = torch.randn(512, cfg["text_dim"], generator=g, device=dev).bfloat16()
    ctx[77:] = 0
    clip_fea = torch.randn(257, 1280, generator=g, device=dev).bfloat16()
    x = torch.randn(3, 36, 60, 104, generator=g, device=dev).bfloat16()
    t = torch.full([3], 611.0, dtype=torch.float32, device=dev)
    fr = [0, 1, 2]
    kc, vc = eng.new_kv_cache(3)
    ck, cv = eng.precompute_context(ctx)
    eng.set_image_kv(*eng.precompute_image_context(clip_fea))
    y = eng.forward(x, t, fr, fr, fr, kc, vc, ck, cv)
    okv = [{n: u.to(dev) for n, u in d.items()} for d in W.new_kv_cache(ocfg, 3, eng.S)]
    yo = W.dit_forward(sd, ocfg, x.permute(1, 0, 2, 3), t.view(1, -1), ctx, okv, [None] * cfg["num_layers"], fr, fr, fr, attn_fn=_grouped_sdpa,
                       clip_fea=clip_fea).permute(1, 0, 2, 3)
    torch.cuda.synchronize()
    e = rel_l2(y, yo)
    print(f"i2v model type, dim 1536 / 480p / 4 layers: rel_l2(HIP, oracle on device) = {e:.3e}")
    assert torch.isfinite(y.float()).all() and e < TOL


def test_full_width_one_layer_vs_cpu_oracle_directly():
    """No device-evaluated hop: Wan2.1-14B WIDTH (dim 5120, 40 heads, ffn 13824) at the 720p geometry, ONE layer, stage s0 (2 query
    frames over 2) and then the anchor stage s1 (7 query frames = 25 200 rows over 9 frames = 32 400 keys) with the live KV cache --
    the large-problem GEMM kernels on the real block shapes, the 64-rows-per-wave attention with 40 heads, 57-tile pages and its
    split-KV tail, the fused epilogues -- against the pinned CPU oracle itself (oracle/wan_dit_ref.py on the host CPU; ~1 min on the
    GPU box's 128 threads).  Measured 4.4e-3 / 4.3e-3 (profiles/r04h_full_width_one_layer_vs_cpu_oracle.log); bound 8e-3 per forward."""
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal
    from oracle import stage_ref
    from oracle import wan_dit_ref as W
    cfg = dict(WAN_CONFIGS["14B"])
    cfg["num_layers"] = 1
    lat = (90, 160)
    sd = dit_state_dict(cfg, seed=33)
    eng = DitEngine(cfg, lat[0], lat[1], "cuda:0")
    eng.load_state_dict(sd)
    ocfg = W.DitCfg(**cfg)
    S = eng.S
    ctx = philox_normal([512, cfg["text_dim"]], 34)
    ctx[77:] = 0
    noise = philox_normal([21, 16, lat[0], lat[1]], 35)
    kc, vc = eng.new_kv_cache(15)
    ck, cv = eng.precompute_context(ctx.cuda())
    okv = W.new_kv_cache(ocfg, 15, S)
    ocross = [None]
    vis = stage_ref.VisIndex()
    for si, frames in enumerate(stage_ref.stage_frames(stage_ref.T2V_CLEAN_STEPS)[:2]):
        vis.on_forward(frames)
        order = vis.slots()
        x = noise[frames].contiguous()
        t = torch.full([len(frames)], (999.0, 750.0)[si], dtype=torch.float32)
        ws = stage_ref.write_slots_for(frames)
        y = eng.forward(x.cuda(), t.cuda(), frames, ws, order, kc, vc, ck, cv).cpu()
        yo = W.dit_forward(sd, ocfg, x.permute(1, 0, 2, 3), t.view(1, -1), ctx, okv, ocross, frames, ws, order).permute(1, 0, 2, 3)
        e = rel_l2(y, yo)
        print(f"14B width, 1 layer, 720p, stage s{si} (Lq {len(frames) * S}, Lkv {len(order) * S}): rel_l2(HIP, CPU oracle) = {e:.3e}")
        assert torch.isfinite(y.float()).all() and e < 8e-3
        slot = [w for w in ws if w >= 0][0]
        assert rel_l2(kc[0, slot * S:(slot + 1) * S].cpu(), okv[0]["k"][0, slot * S:(slot + 1) * S].reshape(S, -1)) < 1e-2
