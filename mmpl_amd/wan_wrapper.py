"""Model wrappers with the reference's names and call shapes (MMPL_t2v/utils/wan_wrapper.py), on the HIP engines.

  * ``WanFPSWrapper``   (:317-515)  generator: forward(noisy, conditional_dict, timestep, kv_cache, crossattn_cache,
                                    current_start, cache_start) -> (flow_pred, pred_x0)
  * ``WanVAEWrapper``   (:54-113)   decode_to_pixel / encode_to_latent
  * ``WanTextEncoder``  (:15-51)    tokenizer -> HIP umT5-xxl engine (mmpl_amd/t5.py) -> padding rows zeroed

``WanFPSWrapper`` also serves the Wan-I2V model type (``config.json`` model_type 'i2v', wan/modules/model.py:563-616):
``forward`` / ``capture`` take ``clip_fea`` and ``y`` (kwargs like ``WanModel.forward``, model.py:626-640, or entries of
``conditional_dict``), build the image K / V once per ``clip_fea`` and concatenate the stage's frames of ``y`` to the
latents on the channel axis (model.py:680-681).

KV caches keep the reference's shape of a list of per-layer dicts (``k``, ``v``, ``attention_vis_index`` ...,
pipeline/casual_fps_inference.py:453-501) so pipeline-style code runs unchanged, but the per-layer tensors are views
into ONE allocation per cache that the HIP forward addresses through its slot table.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from .dit import DitEngine
from .geometry import Geometry
from .scheduler import FlowMatchScheduler
from .stage_plan import StagePlan, slot_of
from .synthetic import WAN_CONFIGS

local_wan_path = "../wan_models"


class KVCache(list):
    """list of per-layer dicts like the reference's kv_cache; `k_all` / `v_all`: [L, n_slots*S, dim]."""

    def __init__(self, engine: DitEngine, n_slots: int = 15):
        super().__init__()
        self.engine = engine
        self.k_all, self.v_all = engine.new_kv_cache(n_slots)
        self.vis: List[int] = []          # token offsets, shared by every layer (the reference keeps L identical copies)
        # what the self-attention's softmax passes did on this branch's previous forward (DitEngine.new_attn_history): travels with
        # the cache because it is per CFG branch; the pipeline zeroes it at every stage boundary.  None = stateless attention.
        self.attn_history: Optional[torch.Tensor] = None
        H = engine.cfg["num_heads"]
        for l in range(engine.L):
            self.append({"k": self.k_all[l].view(1, -1, H, 128), "v": self.v_all[l].view(1, -1, H, 128),
                         "global_end_index": torch.tensor([0], dtype=torch.long),
                         "local_end_index": torch.tensor([0], dtype=torch.long),
                         "attention_vis_index": self.vis})

    def reset(self):
        self.vis.clear()
        self.reset_attn_history()

    def enable_attn_history(self, on: bool = True):
        if on and self.attn_history is None:
            self.attn_history = self.engine.new_attn_history()
        elif not on:
            self.attn_history = None

    def reset_attn_history(self):
        if self.attn_history is not None:
            self.attn_history.zero_()


class CrossAttnCache(list):
    """list of per-layer dicts {"k","v","is_init"}; K/V for all layers are filled by one precompute call."""

    def __init__(self, engine: DitEngine):
        super().__init__()
        self.engine = engine
        self.k_all = torch.zeros(engine.L, engine.text_len, engine.dim, dtype=torch.bfloat16, device=engine.device)
        self.v_all = torch.zeros_like(self.k_all)
        self.rows = engine.text_len       # rows `rows .. text_len-1` of k_all / v_all repeat one row (CrossKV.rows of the contents)
        H = engine.cfg["num_heads"]
        for l in range(engine.L):
            self.append({"k": self.k_all[l].view(1, -1, H, 128), "v": self.v_all[l].view(1, -1, H, 128), "is_init": False})

    @property
    def is_init(self) -> bool:
        return all(b["is_init"] for b in self)

    def fill(self, prompt_embeds: torch.Tensor):
        self.rows = self.engine.precompute_context(prompt_embeds, out=(self.k_all, self.v_all)).rows
        for b in self:
            b["is_init"] = True


class _HeldGraph:
    """A hipGraph together with the buffers it captured BY ADDRESS that nobody else owns (the i2v model type's 36-channel input:
    allocated outside the capture, so it would go back to the allocator -- and be overwritten -- once capture() returns)."""

    def __init__(self, graph, *keep):
        self.graph, self.keep = graph, keep

    def replay(self):
        self.graph.replay()


class _ModelHandle:
    """What the pipeline touches on `generator.model` (num_frame_per_block, parameters())."""

    def __init__(self, engine: DitEngine):
        self.engine = engine
        self.num_frame_per_block = 1

    def parameters(self):
        return iter(self.engine._weights)


class WanFPSWrapper(torch.nn.Module):
    def __init__(self, model_name="Wan2.1-T2V-14B", timestep_shift=8.0, is_causal=False, local_attn_size=-1, sink_size=0,
                 *, model_config: Optional[dict] = None, geometry: Optional[Geometry] = None, device="cuda:0"):
        super().__init__()
        assert is_causal, "only the causal FPS generator is on the hot path"
        self.geometry = geometry or Geometry.named("480p")
        from .checkpoints import read_diffusers_dir
        cfg = model_config
        wdir = f"{local_wan_path}/{model_name}/"
        disk_cfg, disk_sd = read_diffusers_dir(wdir)          # CausalFPSWanModel.from_pretrained(...) (wan_wrapper.py:328-330)
        if cfg is None:
            cfg = disk_cfg
        if cfg is None:
            key = "14B" if "14B" in model_name else "1.3B"
            cfg = WAN_CONFIGS[key]
        self.engine = DitEngine(cfg, self.geometry.lat_h, self.geometry.lat_w, device)
        self.model = _ModelHandle(self.engine)
        if disk_sd is not None:
            self.engine.load_state_dict(disk_sd)
        self.model_type = self.engine.model_type
        self._clip_src = None                 # (clip_fea tensor, its version counter) the engine's image K/V were built from
        self.uniform_timestep = not is_causal
        self.scheduler = FlowMatchScheduler(shift=timestep_shift, sigma_min=0.0, extra_one_step=True)
        self.scheduler.set_timesteps(1000, training=True)
        self.seq_len = self.geometry.frame_seqlen * self.geometry.frames_per_chunk

    # nn.Module-compatible entry points the entry scripts use
    def load_state_dict(self, state_dict, strict: bool = True):
        """MMPL .pt checkpoints hold {'generator': {'model.<key>': tensor}} (Wan_fps_inference_1gpu.py:66-68)."""
        from .checkpoints import strip_generator_prefix
        self.engine.load_state_dict(strip_generator_prefix(state_dict))
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def to(self, *args, **kwargs):
        return self                       # weights already live on the GPU in bf16

    def get_scheduler(self):
        return self.scheduler

    def new_kv_cache(self, n_slots: int = 15) -> KVCache:
        return KVCache(self.engine, n_slots)

    def new_crossattn_cache(self) -> CrossAttnCache:
        return CrossAttnCache(self.engine)

    def _image_stream(self, frames, conditional_dict, clip_fea, y):
        """Wan-I2V model type: make sure the engine's image K/V belong to this `clip_fea` (MLPProj + per-block k_img / v_img,
        once per image) and return the stage's slice of the conditioning video, [nF, 20, h, w] bf16 (None for t2v)."""
        if self.model_type != "i2v":
            return None
        clip_fea = conditional_dict.get("clip_fea") if clip_fea is None else clip_fea
        y = conditional_dict.get("y") if y is None else y
        if clip_fea is None or y is None:
            raise ValueError("WanFPSWrapper: an i2v model needs clip_fea and y (model.py:672-673)")
        key = (clip_fea, clip_fea._version)       # same tensor object AND not written in place since: the K/V are still its
        if self._clip_src is None or self._clip_src[0] is not key[0] or self._clip_src[1] != key[1]:
            fea = clip_fea[0] if clip_fea.dim() == 3 else clip_fea
            self.engine.set_image_kv(*self.engine.precompute_image_context(fea))
            self._clip_src = key
        yy = y[0] if isinstance(y, (list, tuple)) or y.dim() == 5 else y              # [20, F, h, w]
        assert yy.shape[0] == self.engine.in_dim - 16 and yy.shape[2:] == (self.engine.lat_h, self.engine.lat_w), tuple(yy.shape)
        # (a stack of views: no host-built index tensor, so this also runs inside a hipGraph capture)
        return torch.stack([yy[:, f] for f in frames], dim=0).to(device=self.engine.device, dtype=torch.bfloat16)

    def capture(self, noisy_image_or_video, conditional_dict, timestep, kv_cache, crossattn_cache, current_start, out,
                clip_fea=None, y=None):
        """hipGraph of this exact forward (fixed buffers / stage shape); returns the graph, replay() re-runs it on the
        current contents of `noisy_image_or_video`, `timestep` and the caches."""
        assert noisy_image_or_video.is_contiguous() and noisy_image_or_video.dtype == torch.bfloat16
        assert timestep.dtype == torch.float32 and timestep.is_contiguous() and out.is_contiguous()
        S = self.engine.S
        if not crossattn_cache.is_init:
            pe = conditional_dict["prompt_embeds"]
            crossattn_cache.fill(pe[0] if pe.dim() == 3 else pe)
        starts = [int(s) for s in current_start]
        frames = [s // S for s in starts]
        vis = kv_cache.vis
        if 15 * S not in starts:
            for s in starts:
                if s not in vis:
                    vis.append(s)
        x, pre = noisy_image_or_video[0], None
        ys = self._image_stream(frames, conditional_dict, clip_fea, y)
        if ys is not None:                    # static 36-channel input; the graph refreshes its latent channels before the forward
            lat, x = x, torch.cat([x, ys], dim=1).contiguous()
            pre = lambda: x[:, :16].copy_(lat)
        g = self.engine.capture(x, timestep.view(-1), frames, StagePlan.write_slots(frames),
                                [slot_of(o // S) for o in vis], kv_cache.k_all, kv_cache.v_all, crossattn_cache.k_all,
                                crossattn_cache.v_all, out[0], pre=pre, cross_rows=crossattn_cache.rows,
                                attn_history=kv_cache.attn_history)
        return g if ys is None else _HeldGraph(g, x)

    def forward(self, noisy_image_or_video: torch.Tensor, conditional_dict: dict, timestep: torch.Tensor,
                kv_cache: Optional[KVCache] = None, crossattn_cache: Optional[CrossAttnCache] = None,
                current_start=None, classify_mode=False, concat_time_embeddings=False, clean_x=None, aug_t=None,
                cache_start=None, out: Optional[torch.Tensor] = None, return_x0: bool = False, clip_fea=None, y=None,
                workspace: Optional[torch.Tensor] = None, share_out: Optional[torch.Tensor] = None,
                share_in: Optional[torch.Tensor] = None):
        assert kv_cache is not None and crossattn_cache is not None, "the FPS path always runs with caches"
        assert noisy_image_or_video.shape[0] == 1, "batch size 1 (as every reference entry point)"
        S = self.engine.S
        if not crossattn_cache.is_init:                                   # model.py:175-180
            pe = conditional_dict["prompt_embeds"]
            crossattn_cache.fill(pe[0] if pe.dim() == 3 else pe)
        starts = [int(s) for s in current_start]
        frames = [s // S for s in starts]
        vis = kv_cache.vis
        if 15 * S not in starts:                                          # causal_fps_model.py:209,219 / 255
            for s in starts:
                if s not in vis:
                    vis.append(s)
        x = noisy_image_or_video[0]
        if x.dtype != torch.bfloat16 or not x.is_contiguous():
            x = x.to(torch.bfloat16).contiguous()
        lat = x
        ys = self._image_stream(frames, conditional_dict, clip_fea, y)
        if ys is not None:
            x = torch.cat([x, ys], dim=1).contiguous()                     # model.py:680-681
        t = timestep.reshape(-1).to(device=x.device, dtype=torch.float32)
        flow = self.engine.forward(x, t, frames, StagePlan.write_slots(frames), [slot_of(o // S) for o in vis],
                                   kv_cache.k_all, kv_cache.v_all, crossattn_cache.k_all, crossattn_cache.v_all,
                                   out=None if out is None else out[0], cross_rows=crossattn_cache.rows, workspace=workspace,
                                   share_out=share_out, share_in=share_in, attn_history=kv_cache.attn_history)
        flow_pred = flow.unsqueeze(0)
        pred_x0 = None
        if return_x0:                                                     # wan_wrapper.py:373-397 (unused by the pipeline)
            sig = self.scheduler.sigmas.double().to(x.device)
            ts = self.scheduler.timesteps.double().to(x.device)
            tid = torch.argmin((ts.unsqueeze(0) - t.double().unsqueeze(1)).abs(), dim=1)
            pred_x0 = (lat.double() - sig[tid].reshape(-1, 1, 1, 1) * flow.double()).to(flow.dtype).unsqueeze(0)
        return flow_pred, pred_x0


class WanTextEncoder(torch.nn.Module):
    """utils/wan_wrapper.py:15-51 on the HIP umT5 engine (mmpl_amd/t5.py): tokenizer (seq_len 512, whitespace clean,
    tokenizers.py:38-82) -> encoder -> padding rows zeroed.  The engine computes in bf16 (the reference's generate.py
    T5 dtype, wan/configs/shared_config.py; its fps wrapper upcasts the same bf16 checkpoint to fp32 on the CPU -- the
    difference is the bf16 rounding the DiT's text_embedding applies to the context anyway; tests/test_t5_gpu.py).

    ``encode_fn(prompts) -> [B, 512, 4096]`` overrides everything (precomputed embeddings); otherwise pass
    ``state_dict`` or let it load ``models_t5_umt5-xxl-enc-bf16.pth`` from ``local_wan_path``."""

    def __init__(self, encode_fn=None, state_dict: Optional[dict] = None, pretrained_path: Optional[str] = None,
                 tokenizer=None, tokenizer_path: Optional[str] = None, cfg: Optional[dict] = None, text_len: int = 512,
                 device="cuda:0"):
        super().__init__()
        self.encode_fn = encode_fn
        self.text_len = text_len
        self.tokenizer = tokenizer
        self.tokenizer_path = tokenizer_path or f"{local_wan_path}/Wan2.1-T2V-14B/google/umt5-xxl/"
        self.model = None
        if encode_fn is not None:
            return
        path = pretrained_path or f"{local_wan_path}/Wan2.1-T2V-14B/models_t5_umt5-xxl-enc-bf16.pth"
        if state_dict is None and os.path.exists(path):
            from .checkpoints import read_state_dict
            state_dict = read_state_dict(path)
        if state_dict is not None:
            from .checkpoints import infer_t5_config
            from .t5 import T5Engine
            self.model = T5Engine(cfg or infer_t5_config(state_dict), text_len=text_len, device=device)
            self.model.load_state_dict(state_dict)

    def to(self, *args, **kwargs):
        return self

    @staticmethod
    def _clean(text: str) -> str:
        """clean='whitespace' (tokenizers.py:12-21); ftfy.fix_text is applied when ftfy is installed."""
        import html
        import re
        try:
            import ftfy
            text = ftfy.fix_text(text)
        except ImportError:
            pass
        text = html.unescape(html.unescape(text)).strip()
        return re.sub(r"\s+", " ", text).strip()

    def tokenize(self, text_prompts: List[str]):
        if self.tokenizer is None:
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(self.tokenizer_path)
        enc = self.tokenizer([self._clean(p) for p in text_prompts], return_tensors="pt", padding="max_length", truncation=True,
                             max_length=self.text_len, add_special_tokens=True)
        return enc.input_ids, enc.attention_mask

    def forward(self, text_prompts: List[str]) -> dict:
        if self.encode_fn is not None:
            return {"prompt_embeds": self.encode_fn(text_prompts)}
        if self.model is None:
            raise RuntimeError("WanTextEncoder: no umT5 weights (models_t5_umt5-xxl-enc-bf16.pth not found and no state_dict / "
                               "encode_fn given)")
        ids, mask = self.tokenize(text_prompts)
        return {"prompt_embeds": self.model.encode(ids, mask)}      # padding rows zeroed inside mmpl_t5_encode


class SyntheticTextEncoder(WanTextEncoder):
    """Deterministic stand-in used by tests / bench: a prompt hashes to a seed for N(0,1) embeddings, pad rows zeroed."""

    def __init__(self, text_dim=4096, device="cuda:0", n_valid=64):
        super().__init__(encode_fn=self._encode)
        self.text_dim, self.dev, self.n_valid = text_dim, device, n_valid

    def _encode(self, text_prompts: List[str]) -> torch.Tensor:
        import zlib
        from .synthetic import philox_normal
        outs = []
        for p in text_prompts:
            e = philox_normal([512, self.text_dim], zlib.crc32(p.encode("utf-8")))
            e[self.n_valid:] = 0
            outs.append(e)
        return torch.stack(outs).to(self.dev)


class WanVAEWrapper(torch.nn.Module):
    """utils/wan_wrapper.py:54-113 on the HIP Wan 3D-VAE engine (mmpl_amd/vae.py)."""

    mean = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
            0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
    std = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
           3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]

    def __init__(self, geometry: Optional[Geometry] = None, device="cuda:0", state_dict: Optional[dict] = None,
                 pretrained_path: Optional[str] = None):
        super().__init__()
        from .vae import VaeEngine
        self.geometry = geometry or Geometry.named("480p")
        self.model = VaeEngine(self.geometry.lat_h, self.geometry.lat_w, device)
        path = pretrained_path or f"{local_wan_path}/Wan2.1-T2V-14B/Wan2.1_VAE.pth"
        if state_dict is None and os.path.exists(path):
            from .checkpoints import read_state_dict
            state_dict = read_state_dict(path)
        if state_dict is not None:
            self.model.load_state_dict(state_dict)

    def to(self, *args, **kwargs):
        return self

    def encode_to_latent(self, pixel: torch.Tensor) -> torch.Tensor:
        """pixel [B, 3, T, H, W] in [-1, 1] -> latent [B, F, 16, h, w] float32 (normalised mu)."""
        return torch.stack([self.model.encode(u, self.mean, self.std).float() for u in pixel], dim=0)

    def decode_to_pixel(self, latent: torch.Tensor, use_cache: bool = False) -> torch.Tensor:
        """latent [B, F, 16, h, w] -> pixel [B, T, 3, 8h, 8w] float32 clamped to [-1, 1]."""
        return torch.stack([self.model.decode(u, self.mean, self.std).float().clamp_(-1, 1) for u in latent], dim=0)
