"""The Wan-I2V MODEL TYPE through the drop-in seams (-m gpu; VERDICT r2 missing #3, BASELINE configs[4]):
WanFPSWrapper(model_type 'i2v') / CausalFPSInferencePipeline(image_condition=) / `mmpl_amd.cli --i2v --i2v_model`.

Oracle: oracle/stage_ref.run_chunk(clip_fea=, y=) -- the stage loop over oracle/wan_dit_ref.dit_forward(clip_fea=), which is
pinned bit-exactly against the reference's WanModel(model_type='i2v') (tests/golden/dit_i2v_tiny.pt).  The conditioning
(clip_fea from the HIP CLIP tower, y from the HIP VAE encode) is computed once and handed to both sides; those two
engines have their own parity tests (test_i2v_clip_gpu.py, test_vae_gpu.py).  y's mask is checked against the literal
recipe of wan/image2video.py:207-214."""
import types

import pytest
import torch

from tests.util import rel_l2

pytestmark = pytest.mark.gpu
LAT = (16, 24)


def _setup(steps=2):
    from mmpl_amd.geometry import Geometry
    from mmpl_amd.i2v_clip import CLIPVisionTower
    from mmpl_amd.pipeline import CausalFPSInferencePipeline
    from mmpl_amd.synthetic import WAN_CONFIGS, clip_visual_state_dict, dit_i2v_state_dict, philox_normal, vae_state_dict
    from mmpl_amd.wan_wrapper import WanFPSWrapper, WanTextEncoder, WanVAEWrapper
    cfg = dict(WAN_CONFIGS["tiny"], model_type="i2v")
    geo = Geometry(*LAT)
    sd = dit_i2v_state_dict(cfg, seed=12)
    gen = WanFPSWrapper("none", is_causal=True, timestep_shift=5.0, model_config=cfg, geometry=geo, device="cuda:0")
    gen.load_state_dict({"model." + k: v for k, v in sd.items()})
    ctx = {}
    for name, seed, nv in (("pos", 31, 40), ("neg", 32, 12)):
        c = philox_normal([1, 512, cfg["text_dim"]], seed)
        c[:, nv:] = 0
        ctx[name] = c
    enc = WanTextEncoder(lambda prompts: (ctx["neg"] if prompts[0] == "NEG" else ctx["pos"]).cuda())
    vae = WanVAEWrapper(geometry=geo, device="cuda:0", state_dict=vae_state_dict(seed=3))
    clip = CLIPVisionTower(224, 14, 1280, 4, 16, 3)
    clip.load_state_dict(clip_visual_state_dict(1280, 16, 3, 224, 14, seed=4))
    args = types.SimpleNamespace(model_kwargs={}, num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0,
                                 negative_prompt="NEG", independent_first_frame=False, sampling_steps=steps)
    pipe = CausalFPSInferencePipeline(args, "cuda:0", generator=gen, text_encoder=enc, vae=vae, save=None, mode="i2v", geometry=geo)
    return pipe, clip, sd, cfg, ctx


def test_first_frame_mask_is_the_reference_recipe():
    from mmpl_amd.i2v_condition import first_frame_mask
    lat_h, lat_w = 3, 5
    msk = torch.ones(1, 81, lat_h, lat_w)                                       # wan/image2video.py:207-214, literally
    msk[:, 1:] = 0
    msk = torch.concat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
    msk = msk.view(1, msk.shape[1] // 4, 4, lat_h, lat_w)
    msk = msk.transpose(1, 2)[0]
    got = first_frame_mask(21, lat_h, lat_w, "cpu")
    assert got.shape == (4, 21, lat_h, lat_w) and torch.equal(got, msk)
    assert got[:, 0].min() == 1 and got[:, 1:].max() == 0


def test_i2v_model_type_chunk_vs_oracle():
    from mmpl_amd.i2v_condition import build_image_condition
    from mmpl_amd.synthetic import philox_normal
    from oracle import stage_ref
    from oracle import wan_dit_ref as W
    pipe, clip, sd, cfg, ctx = _setup()
    img = philox_normal([3, LAT[0] * 8, LAT[1] * 8], 71).clamp(-1, 1)
    cond = build_image_condition(pipe.vae, clip, img.cuda())
    assert cond["clip_fea"].shape == (257, 1280) and cond["y"].shape == (20, 21, *LAT)
    assert cond["y"][:4, 0].float().min() == 1 and cond["y"][:4, 1:].float().abs().max() == 0
    # the latent half of y is the VAE encode of [image, 80 zero frames] (image2video.py:236-244)
    clipv = torch.zeros(1, 3, 81, LAT[0] * 8, LAT[1] * 8, dtype=torch.bfloat16, device="cuda")
    clipv[0, :, 0] = img.cuda()
    assert torch.equal(cond["y"][4:], pipe.vae.encode_to_latent(clipv)[0].permute(1, 0, 2, 3).to(torch.bfloat16))
    image_latent = pipe.vae.encode_to_latent(img.cuda()[None, :, None]).to(torch.bfloat16)      # [1, 1, 16, h, w]
    noise = philox_normal([1, 21, 16, *LAT], 24)
    got = {}
    pipe.handoff_sink = lambda t: got.setdefault("h", t.clone())
    _, lat = pipe.inference(noise.cuda(), ["a cat"], initial_latent=image_latent, return_latents=True, decode=False, image_condition=cond)
    torch.cuda.synchronize()
    ocfg = W.DitCfg(**{k: v for k, v in cfg.items() if k != "model_type"}, in_dim=36)
    o_out, o_hand, _ = stage_ref.run_chunk(sd, ocfg, noise, ctx["pos"][0], ctx["neg"][0], None, image_latent.cpu(), "i2v", 5.0, 2, 5.0,
                                           gpu_scalar_semantics=True, clip_fea=cond["clip_fea"].cpu(), y=cond["y"].cpu())
    e, eh = rel_l2(lat, o_out), rel_l2(got["h"], o_hand)
    print(f"i2v model type chunk: rel_l2 latents = {e:.3e}, hand-off = {eh:.3e}")
    assert got["h"].shape == (1, 3, 16, *LAT) and e < 1.5e-2 and eh < 1.5e-2
    # graph replay == eager launches, bit for bit (the 36-channel input buffer is refreshed inside the graph)
    pipe.use_graphs = False
    _, lat_eager = pipe.inference(noise.cuda(), ["a cat"], initial_latent=image_latent, return_latents=True, decode=False, image_condition=cond)
    assert torch.equal(lat_eager, lat)
    # the image stream matters: another image -> other latents; and a t2v-style call without it is an error
    pipe.use_graphs = True
    cond2 = build_image_condition(pipe.vae, clip, -img.cuda())
    _, lat2 = pipe.inference(noise.cuda(), ["a cat"], initial_latent=image_latent, return_latents=True, decode=False, image_condition=cond2)
    assert rel_l2(lat2, lat) > 1e-2
    with pytest.raises(ValueError):
        pipe.inference(noise.cuda(), ["a cat"], initial_latent=image_latent, return_latents=True, decode=False)


def test_cli_i2v_model_type_two_chunks(tmp_path):
    """BASELINE configs[4] as literally named, from the CLI with synthetic weights: image -> CLIP tower + VAE -> Wan-I2V
    model type, I2V stage plan, 2 chunks (the second conditioned on the first one's 5th-last frame)."""
    import numpy as np
    from PIL import Image
    from mmpl_amd import cli
    rng = np.random.default_rng(1)
    img = tmp_path / "in.png"
    Image.fromarray(rng.integers(0, 255, (90, 130, 3), dtype=np.uint8)).save(img)
    out = tmp_path / "out"
    cli.main(["--synthetic", "--model", "tiny", "--latent_hw", "16", "24", "--duration", "2", "--sampling_steps", "1", "--i2v", "--i2v_model",
              "--image", str(img), "--output_folder", str(out)])
    v = torch.load(out / "0-0.pt")
    assert v.shape == (81 + 76, 128, 192, 3) and v.dtype == torch.uint8 and v.float().std() > 1.0
