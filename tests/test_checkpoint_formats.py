"""the on-disk formats the reference's entry points read, round-tripped through synthetic files.

CPU part: the host-side readers (mmpl_amd/checkpoints.py).  GPU part (-m gpu): a pipeline constructed with DEFAULT
arguments -- generator / text encoder / VAE all loaded from `local_wan_path` like the reference's constructor does
(pipeline/casual_fps_inference.py:54-69) -- plus the MMPL `.pt` on top, equals the pipeline built from the same tensors
in memory, bit for bit."""
import json
import os
import types

import pytest
import torch

from mmpl_amd import checkpoints
from mmpl_amd.synthetic import T5_CONFIGS, WAN_CONFIGS, dit_state_dict, t5_state_dict, vae_state_dict

LAT = (16, 24)
DIT = dict(WAN_CONFIGS["tiny"], text_dim=T5_CONFIGS["tiny"]["dim"])       # the DiT consumes the text encoder's width


def _write_wan_dir(root, dit_sd, n_shards=3, with_t5=True, name="Wan2.1-T2V-14B", model_type="t2v", in_dim=16):
    """../wan_models/Wan2.1-T2V-14B/ as the reference expects it (tiny tensors under the real names)."""
    from safetensors.torch import save_file
    wdir = os.path.join(root, name)
    os.makedirs(wdir)
    cfg = DIT
    with open(os.path.join(wdir, "config.json"), "w") as f:
        json.dump(dict(_class_name="CausalFPSWanModel", model_type=model_type, patch_size=[1, 2, 2], text_len=512, in_dim=in_dim, out_dim=16,
                       eps=1e-6, **cfg), f)
    keys = list(dit_sd)
    per = (len(keys) + n_shards - 1) // n_shards
    for i in range(n_shards):
        part = {k: dit_sd[k].contiguous() for k in keys[i * per:(i + 1) * per]}
        save_file(part, os.path.join(wdir, f"diffusion_pytorch_model-{i + 1:05d}-of-{n_shards:05d}.safetensors"))
    torch.save(dict(vae_state_dict(seed=3)), os.path.join(wdir, "Wan2.1_VAE.pth"))
    if with_t5:
        torch.save(dict(t5_state_dict(T5_CONFIGS["tiny"], seed=9)), os.path.join(wdir, "models_t5_umt5-xxl-enc-bf16.pth"))
    return wdir


def test_readers_round_trip(tmp_path):
    base = dit_state_dict(DIT, seed=2)
    wdir = _write_wan_dir(str(tmp_path), base)
    cfg, sd = checkpoints.read_diffusers_dir(wdir)
    assert {k: cfg[k] for k in DIT} == DIT and cfg["model_type"] == "t2v" and cfg["in_dim"] == 16 and cfg["text_len"] == 512
    assert list(sorted(sd)) == list(sorted(base)) and all(torch.equal(sd[k], base[k]) for k in base)
    # a missing shard is an error, not a silently partial model
    os.remove(os.path.join(wdir, "diffusion_pytorch_model-00002-of-00003.safetensors"))
    with pytest.raises(FileNotFoundError):
        checkpoints.read_diffusers_dir(wdir)
    assert checkpoints.read_diffusers_dir(str(tmp_path / "nowhere")) == (None, None)
    # MMPL .pt: 'generator' / 'generator_ema' sections, keys prefixed with 'model.'
    ema = dit_state_dict(DIT, seed=5)
    pt = str(tmp_path / "t2v_tiny.pt")
    torch.save({"generator": {"model." + k: v for k, v in base.items()}, "generator_ema": {"model." + k: v for k, v in ema.items()},
                "step": 8000}, pt)
    g = checkpoints.read_mmpl_checkpoint(pt)
    e = checkpoints.read_mmpl_checkpoint(pt, use_ema=True)
    assert all(torch.equal(g[k], base[k]) for k in base) and all(torch.equal(e[k], ema[k]) for k in ema)
    torch.save({"generator": {}}, pt)
    with pytest.raises(KeyError):
        checkpoints.read_mmpl_checkpoint(pt, use_ema=True)
    # flat pickled state dicts + the umT5 dimensions recovered from the tensors
    t5 = checkpoints.read_state_dict(os.path.join(wdir, "models_t5_umt5-xxl-enc-bf16.pth"))
    assert checkpoints.infer_t5_config(t5) == T5_CONFIGS["tiny"]
    assert checkpoints.infer_t5_config(t5_state_dict(T5_CONFIGS["small"], seed=1)) == T5_CONFIGS["small"]
    vae = checkpoints.read_state_dict(os.path.join(wdir, "Wan2.1_VAE.pth"))
    assert list(vae) == list(vae_state_dict(seed=3))


def test_i2v_config_keeps_model_type_and_in_dim(tmp_path):
    """model_type / in_dim used to be dropped, so a Wan-I2V directory built a t2v engine with in_dim 16."""
    from mmpl_amd.synthetic import dit_i2v_state_dict
    sd = dit_i2v_state_dict(dict(DIT, model_type="i2v"), seed=4)
    wdir = _write_wan_dir(str(tmp_path), sd, name="Wan2.1-I2V-14B-720P", model_type="i2v", in_dim=36, with_t5=False)
    cfg, got = checkpoints.read_diffusers_dir(wdir)
    assert cfg["model_type"] == "i2v" and cfg["in_dim"] == 36 and cfg["eps"] == 1e-6
    assert got["patch_embedding.weight"].shape[1] == 36 and "img_emb.proj.1.weight" in got and "blocks.0.cross_attn.k_img.weight" in got
    with open(os.path.join(wdir, "config.json")) as f:
        j = json.load(f)
    j["model_type"] = "flf2v"
    with open(os.path.join(wdir, "config.json"), "w") as f:
        json.dump(j, f)
    with pytest.raises(ValueError):
        checkpoints.read_diffusers_dir(wdir)
    # the CLIP checkpoint: only the vision tower is taken
    from mmpl_amd.synthetic import clip_visual_state_dict
    vis = clip_visual_state_dict(64, 2, 2, 28, 14, seed=1)
    path = str(tmp_path / "models_clip.pth")
    torch.save({**{"visual." + k: v for k, v in vis.items()}, "textual.token_embedding.weight": torch.zeros(4, 4), "log_scale": torch.zeros(())}, path)
    back = checkpoints.read_clip_visual(path)
    assert list(back) == list(vis) and all(torch.equal(back[k], vis[k]) for k in vis)


@pytest.mark.gpu
def test_i2v_diffusers_dir_through_the_wrapper(tmp_path, monkeypatch):
    """A Wan-I2V directory (config.json model_type 'i2v', in_dim 36, img_emb / k_img / v_img tensors) loaded by
    WanFPSWrapper's default path builds an i2v engine that runs and equals the engine loaded from memory."""
    import mmpl_amd.wan_wrapper as ww
    from mmpl_amd.geometry import Geometry
    from mmpl_amd.synthetic import dit_i2v_state_dict, philox_normal
    cfg = dict(DIT, model_type="i2v")
    sd = dit_i2v_state_dict(cfg, seed=4)
    _write_wan_dir(str(tmp_path), sd, name="Wan2.1-I2V-14B-720P", model_type="i2v", in_dim=36, with_t5=False)
    monkeypatch.setattr(ww, "local_wan_path", str(tmp_path))
    geo = Geometry(*LAT)
    outs = []
    for from_disk in (True, False):
        gen = (ww.WanFPSWrapper("Wan2.1-I2V-14B-720P", is_causal=True, geometry=geo, device="cuda:0") if from_disk else
               ww.WanFPSWrapper("none", is_causal=True, model_config=cfg, geometry=geo, device="cuda:0"))
        if not from_disk:
            gen.load_state_dict(sd)
        assert gen.model_type == "i2v" and gen.engine.in_dim == 36
        kv, cross = gen.new_kv_cache(), gen.new_crossattn_cache()
        cond = {"prompt_embeds": philox_normal([1, 512, DIT["text_dim"]], 5).cuda(), "clip_fea": philox_normal([257, 1280], 6).cuda(),
                "y": philox_normal([20, 21, *LAT], 7).cuda()}
        x = philox_normal([1, 2, 16, *LAT], 8).cuda()
        S = geo.frame_seqlen
        flow, _ = gen(x, cond, torch.full([1, 2], 999.0, device="cuda"), kv, cross, current_start=[0, S], cache_start=[0, S])
        outs.append(flow)
        with pytest.raises(ValueError):
            gen(x, {"prompt_embeds": cond["prompt_embeds"]}, torch.full([1, 2], 999.0, device="cuda"), kv, cross, current_start=[0, S])
    assert torch.isfinite(outs[0].float()).all() and outs[0].float().std() > 0 and torch.equal(outs[0], outs[1])


class _Tok:
    """stands in for the umT5 sentencepiece tokenizer directory (not redistributable, not in the image)"""

    def __call__(self, texts, **kw):
        n, L = len(texts), kw["max_length"]
        ids = torch.zeros(n, L, dtype=torch.long)
        mask = torch.zeros(n, L, dtype=torch.long)
        for i, t in enumerate(texts):
            toks = [3 + (ord(c) % 900) for c in t][:L - 1] + [1]
            ids[i, :len(toks)] = torch.tensor(toks)
            mask[i, :len(toks)] = 1
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)


@pytest.mark.gpu
def test_default_constructed_pipeline_loads_everything_from_disk(tmp_path, monkeypatch):
    import mmpl_amd.wan_wrapper as ww
    from mmpl_amd.geometry import Geometry
    from mmpl_amd.pipeline import CausalFPSInferencePipeline
    from mmpl_amd.synthetic import philox_normal
    base = dit_state_dict(DIT, seed=2)
    tuned = dit_state_dict(DIT, seed=7)
    _write_wan_dir(str(tmp_path), base)
    pt = str(tmp_path / "t2v_tiny.pt")
    torch.save({"generator": {"model." + k: v for k, v in tuned.items()}}, pt)
    monkeypatch.setattr(ww, "local_wan_path", str(tmp_path))
    geo = Geometry(*LAT)
    args = types.SimpleNamespace(model_kwargs=dict(timestep_shift=5.0), num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0,
                                 negative_prompt="bad", independent_first_frame=False, sampling_steps=1)
    # generator=None, text_encoder=None, vae=None: the reference's constructor path
    pipe = CausalFPSInferencePipeline(args, "cuda:0", save=None, geometry=geo)
    assert pipe.generator_cond.engine.L == 2 and pipe.text_encoder.model is not None and pipe.vae.model._weights
    pipe.generator_cond.load_state_dict(checkpoints.read_mmpl_checkpoint(pt))           # Wan_fps_inference_1gpu.py:66-68
    pipe.text_encoder.tokenizer = _Tok()
    noise = philox_normal([1, 21, 16, *LAT], 23).cuda()
    torch.manual_seed(77)                      # the stage loop draws fresh noise for the re-noised frames (torch.randn_like)
    video, lat = pipe.inference(noise, ["a cat"], return_latents=True)
    # the same tensors handed over in memory
    gen = ww.WanFPSWrapper(is_causal=True, timestep_shift=5.0, model_config=DIT, geometry=geo, device="cuda:0")
    gen.load_state_dict(tuned)
    enc = ww.WanTextEncoder(state_dict=t5_state_dict(T5_CONFIGS["tiny"], seed=9), cfg=T5_CONFIGS["tiny"], tokenizer=_Tok(), device="cuda:0")
    vae = ww.WanVAEWrapper(geometry=geo, device="cuda:0", state_dict=vae_state_dict(seed=3))
    pipe2 = CausalFPSInferencePipeline(args, "cuda:0", generator=gen, text_encoder=enc, vae=vae, save=None, geometry=geo)
    torch.manual_seed(77)
    video2, lat2 = pipe2.inference(noise, ["a cat"], return_latents=True)
    assert torch.isfinite(lat.float()).all() and torch.equal(lat, lat2) and torch.equal(video, video2)
