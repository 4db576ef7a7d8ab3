"""Readers of the on-disk formats the reference's entry points consume -- host side only, no GPU:

  * Wan diffusers directory: ``config.json`` + ``diffusion_pytorch_model*.safetensors`` (one file or the sharded
    ``-0000x-of-0000y`` set), what ``CausalFPSWanModel.from_pretrained`` reads (MMPL_t2v/utils/wan_wrapper.py:328-330)
  * MMPL checkpoint ``.pt``: ``{'generator': {'model.<key>': tensor}, 'generator_ema': {...}}``
    (Wan_fps_inference_1gpu.py:66-68)
  * ``Wan2.1_VAE.pth`` (wan/modules/vae.py:628-634) and ``models_t5_umt5-xxl-enc-bf16.pth`` (utils/wan_wrapper.py:25-28):
    plain pickled state dicts

Everything returns CPU tensors keyed like the reference modules' ``state_dict()``; the engines repack them for the HIP kernels.
"""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional, Tuple

import torch

_DIT_CFG_KEYS = ("dim", "ffn_dim", "num_heads", "num_layers", "text_dim", "freq_dim")
# optional keys of WanModel's config (wan/modules/model.py:528-545): carried when present so that a Wan-I2V directory
# (model_type 'i2v', in_dim 36) builds an i2v engine -- dropping them used to make load_state_dict die on patch_embedding
_DIT_OPT_KEYS = ("model_type", "in_dim", "out_dim", "eps", "text_len")


def read_diffusers_dir(wdir: str) -> Tuple[Optional[dict], Optional[Dict[str, torch.Tensor]]]:
    """(model config or None, state dict or None) of a Wan diffusers directory; (None, None) if `wdir` has neither."""
    cfg, sd = None, None
    cpath = os.path.join(wdir, "config.json")
    if os.path.isfile(cpath):
        with open(cpath) as f:
            j = json.load(f)
        defaults = dict(text_dim=4096, freq_dim=256)
        cfg = {k: j.get(k, defaults.get(k)) for k in _DIT_CFG_KEYS}
        missing = [k for k, v in cfg.items() if v is None]
        if missing:
            raise ValueError(f"{cpath}: missing model dimensions {missing}")
        cfg.update({k: j[k] for k in _DIT_OPT_KEYS if j.get(k) is not None})
        if cfg.get("model_type", "t2v") not in ("t2v", "i2v"):
            raise ValueError(f"{cpath}: unsupported model_type {cfg['model_type']!r}")
        if cfg.get("out_dim", 16) != 16:
            raise ValueError(f"{cpath}: out_dim {cfg['out_dim']} (the Wan2.1 latent has 16 channels)")
    if os.path.isdir(wdir):
        names = sorted(n for n in os.listdir(wdir) if re.fullmatch(r"diffusion_pytorch_model(-\d+-of-\d+)?\.safetensors", n))
        shards = [n for n in names if "-of-" in n]
        if shards:
            total = {int(re.search(r"-of-(\d+)\.", n).group(1)) for n in shards}
            if len(total) != 1 or len(shards) != total.pop():
                raise FileNotFoundError(f"{wdir}: incomplete safetensors shard set {shards}")
            names = shards
        if names:
            from safetensors.torch import load_file
            sd = {}
            for n in names:
                part = load_file(os.path.join(wdir, n))
                dup = set(part) & set(sd)
                if dup:
                    raise ValueError(f"{wdir}/{n}: keys repeated across shards: {sorted(dup)[:3]}")
                sd.update(part)
    return cfg, sd


def read_clip_visual(path: str) -> Dict[str, torch.Tensor]:
    """The vision tower of ``models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth`` (wan/modules/clip.py:514-516 loads
    the whole XLMRobertaCLIP state dict; Wan-I2V only ever calls ``model.visual``): the ``visual.*`` tensors with the
    prefix removed, i.e. the keys ``CLIPVisionTower.load_state_dict`` takes."""
    sd = read_state_dict(path)
    out = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
    if not out:
        raise ValueError(f"{path}: no 'visual.*' tensors")
    return out


def read_mmpl_checkpoint(path: str, use_ema: bool = False) -> Dict[str, torch.Tensor]:
    """The generator weights of an MMPL ``.pt`` with the ``model.`` prefix of WanFPSWrapper's keys removed."""
    blob = torch.load(path, map_location="cpu", weights_only=False)
    section = "generator_ema" if use_ema else "generator"
    if section not in blob:
        raise KeyError(f"{path}: no '{section}' section (has {sorted(blob)[:4]})")
    return strip_generator_prefix(blob[section])


def strip_generator_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """'model.blocks.0...' -> 'blocks.0...' (also drops an FSDP / compile wrapper prefix in front of it)."""
    out = {}
    for k, v in sd.items():
        k = re.sub(r"^(module\.|_orig_mod\.|_fsdp_wrapped_module\.)+", "", k)
        out[k[len("model."):] if k.startswith("model.") else k] = v
    return out


def read_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """A pickled state dict (`Wan2.1_VAE.pth`, the umT5 `.pth`), CPU tensors."""
    sd = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(sd, dict) or not all(isinstance(v, torch.Tensor) for v in sd.values()):
        raise ValueError(f"{path}: not a flat state dict")
    return sd


def infer_t5_config(sd: Dict[str, torch.Tensor]) -> dict:
    """Encoder dimensions from the tensors of a umT5 state dict (wan/modules/t5.py:267-312 key layout)."""
    vocab, dim = sd["token_embedding.weight"].shape
    layers = 1 + max(int(m.group(1)) for m in (re.match(r"blocks\.(\d+)\.", k) for k in sd) if m)
    buckets, heads = sd["blocks.0.pos_embedding.embedding.weight"].shape
    return dict(vocab=int(vocab), dim=int(dim), dim_attn=int(sd["blocks.0.attn.q.weight"].shape[0]),
                dim_ffn=int(sd["blocks.0.ffn.fc1.weight"].shape[0]), num_heads=int(heads), num_layers=layers, num_buckets=int(buckets))
