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


def _write_wan_dir(root, dit_sd, n_shards=3, with_t5=True):
    """../wan_models/Wan2.1-T2V-14B/ as the reference expects it (tiny tensors under the real names)."""
    from safetensors.torch import save_file
    wdir = os.path.join(root, "Wan2.1-T2V-14B")
    os.makedirs(wdir)
    cfg = DIT
    with open(os.path.join(wdir, "config.json"), "w") as f:
        json.dump(dict(_class_name="CausalFPSWanModel", model_type="t2v", patch_size=[1, 2, 2], text_len=512, in_dim=16, out_dim=16,
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
    assert cfg == DIT
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
