"""End-to-end: CausalFPSInferencePipeline.inference (HIP) vs the oracle's re-enactment of the reference stage loop,
including the hand-off tensor, the chunk >= 2 path (initial_latent), I2V mode and the VAE decode (-m gpu).

Trajectory-level tolerance (stated): rel-L2 <= 1.5e-2 after 4 stages x (2 UniPC steps x 2 CFG forwards + refresh) --
measured 5.6-5.9e-3; the reference itself decorrelates by 5.5e-3 on this test when only its K/V gather ORDER changes
(tests/golden), so the bound is 2.6x the measured value / 2.7x the reference's own order-noise, not a 7x slack."""
import types

import pytest
import torch

from tests.util import max_abs, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
LAT = (16, 24)
TRAJ_TOL = 1.5e-2


def _setup(mode="t2v", steps=2, with_vae=False, lat=LAT, cfg_name="tiny", weight_seed=2, ctx_seeds=(21, 22), n_valid=(40, 12), weights_on="cpu"):
    from mmpl_amd.geometry import Geometry
    from mmpl_amd.pipeline import CausalFPSInferencePipeline
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, philox_normal, vae_state_dict
    from mmpl_amd.wan_wrapper import WanFPSWrapper, WanTextEncoder, WanVAEWrapper
    cfg = WAN_CONFIGS[cfg_name]
    geo = Geometry(*lat)
    sd = dit_state_dict(cfg, seed=weight_seed, device=weights_on)      # ("cuda:0": 14B-sized weights are drawn on the device)
    gen = WanFPSWrapper(is_causal=True, timestep_shift=5.0, model_config=cfg, geometry=geo, device="cuda:0")
    gen.load_state_dict({"model." + k: v for k, v in sd.items()})          # MMPL checkpoint key style
    ctx = {}
    for name, seed, nv in (("pos", ctx_seeds[0], n_valid[0]), ("neg", ctx_seeds[1], n_valid[1])):
        c = philox_normal([1, 512, cfg["text_dim"]], seed)
        c[:, nv:] = 0
        ctx[name] = c
    enc = WanTextEncoder(lambda prompts: (ctx["neg"] if prompts[0] == "NEG" else ctx["pos"]).cuda())
    vsd = vae_state_dict(seed=3)
    vae = WanVAEWrapper(geometry=geo, device="cuda:0", state_dict=vsd) if with_vae else types.SimpleNamespace()
    args = types.SimpleNamespace(model_kwargs={}, num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0,
                                 negative_prompt="NEG", independent_first_frame=False, sampling_steps=steps)
    pipe = CausalFPSInferencePipeline(args, "cuda:0", generator=gen, text_encoder=enc, vae=vae, save=None, mode=mode, geometry=geo)
    return pipe, sd, vsd, cfg, ctx


def _oracle(sd, cfg, ctx, noise, renoise, initial, mode, steps):
    from oracle import stage_ref
    from oracle import wan_dit_ref as W
    return stage_ref.run_chunk(sd, W.DitCfg(**cfg), noise, ctx["pos"][0], ctx["neg"][0], renoise, initial, mode, 5.0, steps, 5.0,
                               gpu_scalar_semantics=True)


def test_t2v_first_chunk_latents_and_handoff():
    from mmpl_amd.synthetic import philox_normal
    pipe, sd, _, cfg, ctx = _setup("t2v")
    noise = philox_normal([1, 21, 16, *LAT], 23)
    renoise = {f: philox_normal([1, 16, *LAT], 100 + f) for f in (4, 9, 13, 18)}
    got = {}
    pipe.handoff_sink = lambda t: got.setdefault("h", t.clone())
    pipe.renoise_override = {k: v.cuda() for k, v in renoise.items()}
    _, lat = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
    torch.cuda.synchronize()
    o_out, o_hand, _ = _oracle(sd, cfg, ctx, noise, renoise, None, "t2v", 2)
    e, eh = rel_l2(lat, o_out), rel_l2(got["h"], o_hand)
    print(f"t2v chunk 1: rel_l2 latents = {e:.3e}, hand-off = {eh:.3e}")
    assert got["h"].shape == (1, 8, 16, *LAT)
    assert e < TRAJ_TOL and eh < TRAJ_TOL
    # hipGraph replay (default) and eager launches give bit-identical chunks
    pipe.use_graphs = False
    _, lat_eager = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
    assert torch.equal(lat_eager, lat)
    pipe.use_graphs = True
    # second call on the same pipeline re-uses (and resets) the caches: chunk >= 2 with an initial latent
    init = philox_normal([1, 2, 16, *LAT], 55)
    _, lat2 = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
    o2, _, _ = _oracle(sd, cfg, ctx, noise, renoise, init, "t2v", 2)
    e2 = rel_l2(lat2, o2)
    print(f"t2v chunk 2 (initial_latent): rel_l2 = {e2:.3e}")
    assert torch.equal(lat2[:, :2].cpu(), init) and e2 < TRAJ_TOL


def test_i2v_chunk_with_image_latent():
    from mmpl_amd.synthetic import philox_normal
    pipe, sd, _, cfg, ctx = _setup("i2v")
    noise = philox_normal([1, 21, 16, *LAT], 24)
    img = philox_normal([1, 1, 16, *LAT], 56)
    got = {}
    pipe.handoff_sink = lambda t: got.setdefault("h", t.clone())
    _, lat = pipe.inference(noise.cuda(), ["a cat"], initial_latent=img.cuda(), return_latents=True, decode=False)
    o_out, o_hand, _ = _oracle(sd, cfg, ctx, noise, None, img, "i2v", 2)
    e = rel_l2(lat, o_out)
    print(f"i2v: rel_l2 latents = {e:.3e}, hand-off = {rel_l2(got['h'], o_hand):.3e}")
    assert got["h"].shape == (1, 3, 16, *LAT) and e < TRAJ_TOL


def test_decode_and_handoff_transform():
    from mmpl_amd.handoff import handoff_to_initial_latent
    from mmpl_amd.synthetic import philox_normal
    from oracle import stage_ref, vae_ref
    pipe, sd, vsd, cfg, ctx = _setup("t2v", steps=1, with_vae=True)
    lat = philox_normal([1, 21, 16, *LAT], 60)
    video = pipe.vae.decode_to_pixel(lat.cuda())
    assert video.shape == (1, 81, 3, LAT[0] * 8, LAT[1] * 8) and video.dtype == torch.float32
    ref = vae_ref.decode_to_pixel(vsd, lat[:, :3], pipe.vae.mean, pipe.vae.std)          # causal prefix: 9 frames
    assert rel_l2(video[:, :9], ref) < 3e-2
    # consumer transform vs the reference recipe (21-latent decode, 81-frame encode) evaluated by the oracle on the prefix
    recv = philox_normal([1, 8, 16, *LAT], 61)
    init = handoff_to_initial_latent(pipe.vae, recv.cuda())
    m = stage_ref.handoff_to_mask_latents(recv)[:, :4]
    px = vae_ref.decode_to_pixel(vsd, m, pipe.vae.mean, pipe.vae.std).to(BF)
    px = (px * 0.5 + 0.5).clamp(0, 1).to(BF)
    clip = (px[:, 8:13] * 2.0 - 1.0).permute(0, 2, 1, 3, 4)
    ref_init = vae_ref.encode_to_latent(vsd, clip, pipe.vae.mean, pipe.vae.std)[:, :2].to(BF)
    e = rel_l2(init, ref_init)
    print(f"hand-off transform: rel_l2 = {e:.3e}")
    assert init.shape == (1, 2, 16, *LAT) and e < 4e-2


def test_cli_two_chunk_rollout_synthetic(tmp_path):
    """The entry point (reference CLI) end to end on one GPU: 2 chunks, tiny model, 16x24 latents, 2 sampling steps:
    rolling hand-over through the VAE, 5-frame overlap dropped when stitching (Wan_fps_inference_1gpu.py:164-203)."""
    from mmpl_amd import cli
    out = tmp_path / "out"
    cli.main(["--synthetic", "--model", "tiny", "--latent_hw", "16", "24", "--duration", "2", "--sampling_steps", "2",
              "--output_folder", str(out)])
    v = torch.load(out / "0-0.pt")
    assert v.shape == (81 + 76, 128, 192, 3) and v.dtype == torch.uint8
    assert v.float().std() > 1.0


def test_cli_i2v_synthetic(tmp_path):
    """--i2v: image -> VAE encode -> latent frame 0 -> I2V stage plan (MMPL_i2v entry path), one chunk."""
    import numpy as np
    from PIL import Image
    from mmpl_amd import cli
    rng = np.random.default_rng(0)
    img = tmp_path / "in.png"
    Image.fromarray(rng.integers(0, 255, (90, 130, 3), dtype=np.uint8)).save(img)
    out = tmp_path / "out"
    cli.main(["--synthetic", "--model", "tiny", "--latent_hw", "16", "24", "--duration", "1", "--sampling_steps", "1", "--i2v",
              "--image", str(img), "--output_folder", str(out)])
    v = torch.load(out / "0-0.pt")
    assert v.shape == (81, 128, 192, 3)


def test_concurrent_cfg_branches_are_bit_identical_to_sequential():
    """Round 5: the cond and the uncond forward of a denoise step as two PARALLEL branches of the step hipGraph (second stream, private
    workspace; where mmpl_amd.stage_plan.concurrent_cfg_pays says so -- every stage of this small model).  Same kernels, same
    arguments: the chunk must not change by a bit, first chunk and chunk >= 2."""
    from mmpl_amd.synthetic import philox_normal
    from mmpl_amd.stage_plan import concurrent_cfg_pays
    pipe, *_ = _setup("t2v", steps=3)
    assert pipe.concurrent_cfg is None and concurrent_cfg_pays(7 * pipe.frame_seq_length, pipe.generator_cond.engine.dim)
    assert not concurrent_cfg_pays(7 * 3600, 5120) and concurrent_cfg_pays(7 * 1560, 1536)        # 14B / 720p anchors: no; 1.3B / 480p: yes
    noise = philox_normal([1, 21, 16, *LAT], 23)
    renoise = {f: philox_normal([1, 16, *LAT], 100 + f) for f in (4, 9, 13, 18)}
    pipe.renoise_override = {k: v.cuda() for k, v in renoise.items()}
    init = philox_normal([1, 2, 16, *LAT], 55)
    res = {}
    for mode in (False, True, None):
        pipe.concurrent_cfg = mode
        _, a = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
        _, b = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
        torch.cuda.synchronize()
        res[mode] = (a.clone(), b.clone())
    assert pipe._side_stream is not None
    for mode in (True, None):
        assert torch.equal(res[mode][0], res[False][0]) and torch.equal(res[mode][1], res[False][1]), mode
    # ... and the sequential step graphs with block 0's self-attention computed once per step (share_block0, the default) against
    # both branches computing it: the same bits again
    pipe.concurrent_cfg, pipe.share_block0 = False, False
    _, a = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
    _, b = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
    torch.cuda.synchronize()
    assert torch.equal(a, res[False][0]) and torch.equal(b, res[False][1])


def test_attention_history_changes_nothing_where_no_block_fails():
    """`pipeline.attn_history` (default True): each CFG branch's KV cache carries the self-attention's pass history from step to step.
    A block whose FAST pass never fails never leaves the zero state, i.e. computes the stateless kernel's bits -- so on weights whose
    scores the FAST pass holds (these) a whole chunk, first and later, is bit-identical with and without it; and the history stays zero."""
    from mmpl_amd.synthetic import philox_normal
    pipe, *_ = _setup("t2v", steps=3)
    noise = philox_normal([1, 21, 16, *LAT], 23)
    renoise = {f: philox_normal([1, 16, *LAT], 100 + f) for f in (4, 9, 13, 18)}
    pipe.renoise_override = {k: v.cuda() for k, v in renoise.items()}
    init = philox_normal([1, 2, 16, *LAT], 55)
    res = {}
    for on in (True, False):
        pipe.attn_history = on
        _, a = pipe.inference(noise.cuda(), ["a cat"], return_latents=True, decode=False)
        _, b = pipe.inference(noise.cuda(), ["a cat"], initial_latent=init.cuda(), return_latents=True, decode=False)
        torch.cuda.synchronize()
        res[on] = (a.clone(), b.clone())
        for kv in (pipe.kv_cache_pos, pipe.kv_cache_neg):
            assert (kv.attn_history is not None) == on
            if on:
                assert int(kv.attn_history.max()) == 0
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
