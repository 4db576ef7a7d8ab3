"""Host-side handle of the HIP Wan DiT forward (libmmpl_hip.so: mmpl_dit_*).

Mirrors what ``CausalFPSWanModel`` (MMPL_t2v/wan/modules/causal_fps_model.py:398-530, 708-837) offers the
wrapper: construct from the model dims, ``load_state_dict`` with the reference's keys, run one inference
forward against a per-layer KV cache.  PyTorch only owns the device memory here; all compute is in the library.

``model_type='i2v'`` (wan/modules/model.py:563-616: in_dim 36, ``img_emb`` = MLPProj(1280, dim), every block's
cross-attention a ``WanI2VCrossAttention``) adds the image stream: ``precompute_image_context(clip_fea)`` builds the
per-layer image K / V once per image, ``set_image_kv`` attaches them, and ``forward`` takes x with the conditioning video
``y`` concatenated on the channel axis (model.py:680-681).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib

_GLOBAL_KEYS = ["patch_embedding.weight", "patch_embedding.bias", "text_embedding.0.weight", "text_embedding.0.bias",
                "text_embedding.2.weight", "text_embedding.2.bias", "time_embedding.0.weight", "time_embedding.0.bias",
                "time_embedding.2.weight", "time_embedding.2.bias", "time_projection.1.weight", "time_projection.1.bias",
                "head.modulation", "head.head.weight", "head.head.bias"]


class CrossKV(tuple):
    """(cross_k, cross_v) of one prompt, each [num_layers, text_len, dim], plus `rows`: rows `rows .. text_len-1` of every layer are
    copies of row `rows` (the zero-padded tail of the text context after the text embedding; `text_len` = no repeated tail).
    Unpacks like the plain pair; `rows` is a statement about the CONTENTS, so hand it on with any copy of them
    (`DitEngine.forward(..., cross_rows=kv.rows)`); a forward that is not given it attends over all text_len rows."""

    def __new__(cls, k: torch.Tensor, v: torch.Tensor, rows: int):
        self = super().__new__(cls, (k, v))
        self.rows = int(rows)
        return self


class DitEngine:
    def __init__(self, cfg: dict, lat_h: int, lat_w: int, device="cuda:0", max_frames: int = 7):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:      # 'cuda' != 'cuda:0' for torch.device.__eq__: normalise once
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lat_h, self.lat_w = lat_h, lat_w
        self.S = (lat_h // 2) * (lat_w // 2)
        self.dim, self.L = cfg["dim"], cfg["num_layers"]
        self.text_len = cfg.get("text_len", 512)
        self.text_dim = cfg.get("text_dim", 4096)
        self.max_frames = max_frames
        self.model_type = cfg.get("model_type", "t2v")
        assert self.model_type in ("t2v", "i2v")
        self.in_dim = cfg.get("in_dim", 36 if self.model_type == "i2v" else 16)
        self.clip_dim = cfg.get("clip_dim", 1280)
        lib = _lib.load()
        self._lib = lib
        c = _lib.MmplDitConfig(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                               num_layers=cfg["num_layers"], text_dim=self.text_dim, freq_dim=cfg.get("freq_dim", 256),
                               in_dim=self.in_dim, out_dim=16, text_len=self.text_len, eps=cfg.get("eps", 1e-6), lat_h=lat_h,
                               lat_w=lat_w, max_frames=max_frames)
        self._c = c
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.mmpl_dit_create(C.byref(c), C.byref(h)), "mmpl_dit_create")
        self._h = h
        self._weights: List[torch.Tensor] = []
        self._ws: Dict[int, torch.Tensor] = {}
        self._ws2: Optional[torch.Tensor] = None
        self._share: Optional[torch.Tensor] = None
        self._ctx_ws: Optional[torch.Tensor] = None
        self._i2v_w: Optional[dict] = None                 # img_emb + per-layer k_img / v_img / norm_k_img (model_type 'i2v')
        self._img_kv: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self._attn_stats: Optional[torch.Tensor] = None
        self._warm = set()                                 # stage shapes (n_frames) that ran eagerly once: capture()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mmpl_dit_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "") -> None:
        """sd uses the reference's CausalFPSWanModel keys (optionally prefixed, e.g. 'model.' in MMPL .pt files,
        Wan_fps_inference_1gpu.py:66-68).  q/k/v are packed into one [3*dim, dim] operand."""
        dev, bf = self.device, torch.bfloat16

        def g(k):
            return sd[prefix + k].to(device=dev, dtype=bf).contiguous()

        w: List[torch.Tensor] = []
        for k in _GLOBAL_KEYS:
            t = g(k)
            if k == "patch_embedding.weight":
                t = t.reshape(t.shape[0], -1)
                pe_k = (t.shape[1] + 63) // 64 * 64         # K of the patch GEMM: 4 * in_dim padded to a multiple of 64 (zeros)
                assert t.shape[1] == 4 * self.in_dim, (tuple(t.shape), self.in_dim)
                t = torch.nn.functional.pad(t, (0, pe_k - t.shape[1])).contiguous()
            w.append(t)
        w.append(torch.stack([g(f"blocks.{i}.modulation").reshape(6, self.dim) for i in range(self.L)]).contiguous())
        for i in range(self.L):
            p = f"blocks.{i}."
            w.append(torch.cat([g(p + "self_attn.q.weight"), g(p + "self_attn.k.weight"), g(p + "self_attn.v.weight")]).contiguous())
            w.append(torch.cat([g(p + "self_attn.q.bias"), g(p + "self_attn.k.bias"), g(p + "self_attn.v.bias")]).contiguous())
            for k in ("self_attn.norm_q.weight", "self_attn.norm_k.weight", "self_attn.o.weight", "self_attn.o.bias",
                      "norm3.weight", "norm3.bias", "cross_attn.q.weight", "cross_attn.q.bias", "cross_attn.norm_q.weight",
                      "cross_attn.k.weight", "cross_attn.k.bias", "cross_attn.norm_k.weight", "cross_attn.v.weight",
                      "cross_attn.v.bias", "cross_attn.o.weight", "cross_attn.o.bias", "ffn.0.weight", "ffn.0.bias",
                      "ffn.2.weight", "ffn.2.bias"):
                w.append(g(p + k))
        n = self._lib.mmpl_dit_num_weights(C.byref(self._c))
        assert len(w) == n, (len(w), n)
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in w])
        _lib.check(self._lib.mmpl_dit_bind_weights(self._h, arr, n), "mmpl_dit_bind_weights")
        self._weights = w      # keep alive: the library borrows the pointers
        if self.model_type == "i2v":
            self._i2v_w = dict(
                img_emb=[g("img_emb." + k) for k in ("proj.0.weight", "proj.0.bias", "proj.1.weight", "proj.1.bias", "proj.3.weight",
                                                     "proj.3.bias", "proj.4.weight", "proj.4.bias")],
                layers=[[g(f"blocks.{i}.cross_attn." + k) for k in ("k_img.weight", "k_img.bias", "v_img.weight", "v_img.bias",
                                                                     "norm_k_img.weight")] for i in range(self.L)])

    # ------------------------------------------------------------------ diagnostics
    def enable_attn_stats(self) -> torch.Tensor:
        """Count the self-attention kernel's query blocks and what its data-dependent softmax passes did with them: int64 [5] on the
        device = {blocks, blocks whose max-free FAST pass failed and that the GENERAL pass redid (both paid), waves (64 rows) that
        held a failing row, blocks their history sent straight to the GENERAL pass, blocks whose FAST pass held on the references
        their history remembered}, incremented by every later forward.
        The counter's address is a kernel argument: a hipGraph captured while stats are on keeps counting on every replay
        whatever `disable_attn_stats` says later, and one captured while they are off never counts.  One atomic per 256-row
        block; `bench.py` switches it on for its diagnostic modes only."""
        if self._attn_stats is None:
            self._attn_stats = torch.zeros(5, dtype=torch.int64, device=self.device)
        _lib.check(self._lib.mmpl_dit_set_attn_stats(self._h, _lib.ptr(self._attn_stats)), "mmpl_dit_set_attn_stats")
        return self._attn_stats

    def disable_attn_stats(self) -> None:
        """Later eager forwards and later captures stop counting (graphs captured before keep their setting)."""
        _lib.check(self._lib.mmpl_dit_set_attn_stats(self._h, None), "mmpl_dit_set_attn_stats")

    def read_attn_stats(self, reset: bool = False) -> Tuple[int, int, int, int, int]:
        """(blocks run, blocks redone, waves that held a failing row, blocks sent straight to GENERAL, blocks whose FAST pass held on
        remembered references) since the counters were last zeroed."""
        if self._attn_stats is None:
            raise RuntimeError("DitEngine.read_attn_stats: enable_attn_stats() was never called on this engine")
        vals = tuple(int(v) for v in self._attn_stats.cpu())
        if reset:
            self._attn_stats.zero_()
        return vals

    def new_attn_history(self, n_frames: Optional[int] = None) -> torch.Tensor:
        """Zeroed history of the self-attention's softmax passes for `forward(attn_history=...)`: a state byte and 128 int16 lane
        references per (layer, head, 256-row query block, split part), sized for stages of up to `n_frames` frames (default: the
        engine's largest; ~160 MB for Wan 14B at 720p).  One buffer per (CFG branch, stage); `zero_()` it when the stage changes
        (include/mmpl_hip.h)."""
        n = self._lib.mmpl_dit_attn_history_bytes(self._h, self.max_frames if n_frames is None else n_frames)
        return torch.zeros(n, dtype=torch.uint8, device=self.device)

    def share_check_failures(self) -> int:
        """MMPL_CHECK_SHARE=1: share_in forwards since the last call whose layer-0 K / V differed from their share_out forward's
        (synchronises; 0 when the switch is off)."""
        n = C.c_longlong(0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mmpl_dit_share_check_failures(self._h, C.byref(n), _lib.stream_ptr()), "mmpl_dit_share_check_failures")
        return n.value

    # ------------------------------------------------------------------ caches
    def new_kv_cache(self, n_slots: int = 15) -> Tuple[torch.Tensor, torch.Tensor]:
        """[num_layers, n_slots*S, dim] K and V; layer l is the reference's kv_cache[l]['k'] viewed [1, n_slots*S, H, 128]
        (casual_fps_inference.py:453-480)."""
        shape = (self.L, n_slots * self.S, self.dim)
        return (torch.zeros(shape, dtype=torch.bfloat16, device=self.device),
                torch.zeros(shape, dtype=torch.bfloat16, device=self.device))

    def precompute_context(self, context: torch.Tensor, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> CrossKV:
        """context: [L<=text_len, text_dim] -> per-layer cross-attention K, V [num_layers, text_len, dim] as a `CrossKV` (unpacks as
        (K, V); `.rows` = the leading rows that are distinct, see CrossKV).  `out`: write into these two buffers instead of new ones."""
        ctx = torch.zeros(self.text_len, self.text_dim, dtype=torch.bfloat16, device=self.device)
        ctx[:context.shape[0]] = context.to(device=self.device, dtype=torch.bfloat16)
        if out is None:
            ck = torch.empty(self.L, self.text_len, self.dim, dtype=torch.bfloat16, device=self.device)
            cv = torch.empty_like(ck)
        else:
            ck, cv = out
            for t in (ck, cv):
                assert t.shape == (self.L, self.text_len, self.dim) and t.dtype == torch.bfloat16 and t.is_contiguous() and t.device == self.device
        if self._ctx_ws is None:
            self._ctx_ws = torch.empty(self._lib.mmpl_dit_context_workspace_bytes(self._h), dtype=torch.uint8, device=self.device)
        rows = C.c_int(self.text_len)
        with torch.cuda.device(self.device):          # the stream handed to the library is this device's current stream
            _lib.check(self._lib.mmpl_dit_precompute_context(self._h, _lib.ptr(ctx), _lib.ptr(ck), _lib.ptr(cv), _lib.ptr(self._ctx_ws),
                                                             self._ctx_ws.numel(), C.byref(rows), _lib.stream_ptr()), "precompute_context")
        return CrossKV(ck, cv, rows.value)

    def precompute_image_context(self, clip_fea: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """clip_fea: [257, clip_dim] (CLIP ViT-H penultimate-block tokens, clip.py:541) -> per-layer image (K, V)
        [num_layers, 257, dim]: img_emb (MLPProj, model.py:469-481), then k_img / norm_k_img and v_img of every block
        (model.py:251-252).  Once per image, like precompute_context once per prompt."""
        if self._i2v_w is None:
            raise RuntimeError("DitEngine: not an i2v model (or weights not loaded)")
        lib, d = self._lib, self.dim
        fea = clip_fea.to(device=self.device, dtype=torch.bfloat16).contiguous()
        n = fea.shape[0]
        assert fea.shape == (n, self.clip_dim)
        ctx_img = torch.empty(n, d, dtype=torch.bfloat16, device=self.device)
        ws = torch.empty(lib.mmpl_i2v_img_proj_workspace_bytes(n, self.clip_dim, d), dtype=torch.uint8, device=self.device)
        arr = (C.c_void_p * 8)(*[t.data_ptr() for t in self._i2v_w["img_emb"]])
        img_k = torch.empty(self.L, n, d, dtype=torch.bfloat16, device=self.device)
        img_v = torch.empty_like(img_k)
        with torch.cuda.device(self.device):
            _lib.check(lib.mmpl_i2v_img_proj(_lib.ptr(fea), n, self.clip_dim, d, arr, _lib.ptr(ctx_img), _lib.ptr(ws), ws.numel(),
                                             _lib.stream_ptr()), "mmpl_i2v_img_proj")
            for l, (kw, kb, vw, vb, nk) in enumerate(self._i2v_w["layers"]):
                _lib.check(lib.mmpl_i2v_img_kv(_lib.ptr(ctx_img), n, d, _lib.ptr(kw), _lib.ptr(kb), _lib.ptr(vw), _lib.ptr(vb), _lib.ptr(nk),
                                               self.cfg.get("eps", 1e-6), _lib.ptr(img_k[l]), _lib.ptr(img_v[l]), _lib.stream_ptr()),
                           "mmpl_i2v_img_kv")
        return img_k, img_v

    def set_image_kv(self, img_k: Optional[torch.Tensor], img_v: Optional[torch.Tensor]) -> None:
        """Attach (or, with None, detach) the per-layer image K / V that every later forward's cross-attention also attends to."""
        if img_k is None:
            _lib.check(self._lib.mmpl_dit_set_image_kv(self._h, None, None, 0), "mmpl_dit_set_image_kv")
            self._img_kv = None
            return
        assert img_k.shape == img_v.shape and img_k.shape[0] == self.L and img_k.shape[2] == self.dim and img_k.is_contiguous()
        _lib.check(self._lib.mmpl_dit_set_image_kv(self._h, _lib.ptr(img_k), _lib.ptr(img_v), img_k.shape[1]), "mmpl_dit_set_image_kv")
        self._img_kv = (img_k, img_v)      # keep alive: the library borrows the pointers

    def workspace(self, n_frames: int) -> torch.Tensor:
        if n_frames not in self._ws:
            nbytes = self._lib.mmpl_dit_workspace_bytes(self._h, n_frames)
            # one buffer sized for the largest stage serves all stages
            big = max(self._ws.values(), key=lambda t: t.numel(), default=None)
            if big is not None and big.numel() >= nbytes:
                self._ws[n_frames] = big
            else:
                self._ws[n_frames] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws[n_frames]

    def shared_block0_buffer(self, n_frames: int) -> torch.Tensor:
        """[n_frames * S, dim] bf16: x after block 0's self-attention residual, handed from the cond forward of a denoise step
        (`share_out`) to the uncond one (`share_in`).  One buffer, sized for the largest stage seen so far."""
        n = n_frames * self.S * self.dim
        if self._share is None or self._share.numel() < n:
            self._share = torch.empty(n, dtype=torch.bfloat16, device=self.device)
        return self._share

    def second_workspace(self, n_frames: int) -> torch.Tensor:
        """A private scratch buffer for a forward that runs CONCURRENTLY with another one of this engine (the uncond branch of a
        denoise step on a second stream): one buffer, sized for the largest stage seen so far, allocated outside any capture."""
        nbytes = self._lib.mmpl_dit_workspace_bytes(self._h, n_frames)
        if self._ws2 is None or self._ws2.numel() < nbytes:
            self._ws2 = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws2

    # ------------------------------------------------------------------ forward
    def forward(self, x: torch.Tensor, t: torch.Tensor, frame_ids: Sequence[int], write_slots: Sequence[int],
                visible_slots: Sequence[int], k_cache: torch.Tensor, v_cache: torch.Tensor, cross_k: torch.Tensor,
                cross_v: torch.Tensor, out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None,
                cross_rows: Optional[int] = None, share_out: Optional[torch.Tensor] = None,
                share_in: Optional[torch.Tensor] = None, attn_history: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [nF, in_dim, lat_h, lat_w] bf16 (i2v: x and y concatenated on the channel axis); t: [nF] float32 (device).
        Returns the flow prediction [nF, 16, lat_h, lat_w].
        `cross_rows`: the `CrossKV.rows` that belongs to the CONTENTS of cross_k / cross_v (rows cross_rows .. text_len-1 repeat one
        row): the text cross-attention then runs over cross_rows + 1 keys.  None = attend over all text_len rows.
        `share_out` / `share_in` ([nF * S, dim] bf16, see `shared_block0_buffer`): the two branches of classifier-free guidance run
        block 0's self-attention on identical inputs; the first forward leaves x after that residual in `share_out`, the second
        takes it as `share_in` and skips the attention and its output projection (include/mmpl_hip.h; bit-identical).
        `attn_history` (`new_attn_history()`): what the self-attention's softmax passes did on the previous forward of this (CFG
        branch, stage); blocks whose FAST pass failed then start in the GENERAL pass.  The output bits then depend on the forwards
        before (include/mmpl_hip.h); None = stateless.
        `workspace`: a private scratch buffer (>= workspace_bytes(nF)) for a forward that runs concurrently with another
        one on a different stream (cond / uncond); default: the engine's own."""
        nF = x.shape[0]
        assert x.is_contiguous() and x.dtype == torch.bfloat16 and x.shape[1:] == (self.in_dim, self.lat_h, self.lat_w)
        assert t.dtype == torch.float32 and t.numel() == nF and t.is_cuda
        if self.model_type == "i2v" and self._img_kv is None:
            raise RuntimeError("DitEngine: an i2v model needs set_image_kv(*precompute_image_context(clip_fea)) before forward")
        for sh in (share_out, share_in):
            assert sh is None or (sh.dtype == torch.bfloat16 and sh.is_contiguous() and sh.numel() >= nF * self.S * self.dim and sh.device == x.device)
        if attn_history is not None:
            need = self._lib.mmpl_dit_attn_history_bytes(self._h, nF)
            assert attn_history.dtype == torch.uint8 and attn_history.is_contiguous() and attn_history.numel() >= need and attn_history.device == x.device
        if out is None:
            out = torch.empty(nF, 16, self.lat_h, self.lat_w, dtype=torch.bfloat16, device=x.device)
        ws = self.workspace(nF) if workspace is None else workspace
        n_slots = k_cache.shape[1] // self.S
        ia = lambda v: (C.c_int * len(v))(*[int(i) for i in v])
        with torch.cuda.device(self.device):          # the stream handed to the library is this device's current stream
            _lib.check(self._lib.mmpl_dit_forward(
                self._h, _lib.ptr(x), _lib.ptr(t), nF, ia(frame_ids), ia(write_slots), ia(visible_slots), len(visible_slots),
                _lib.ptr(k_cache), _lib.ptr(v_cache), n_slots, _lib.ptr(cross_k), _lib.ptr(cross_v),
                self.text_len if cross_rows is None else int(cross_rows),
                None if share_out is None else _lib.ptr(share_out), None if share_in is None else _lib.ptr(share_in),
                None if attn_history is None else _lib.ptr(attn_history), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "mmpl_dit_forward")
        return out

    # ------------------------------------------------------------------ hipGraph
    def capture(self, x: torch.Tensor, t: torch.Tensor, frame_ids, write_slots, visible_slots, k_cache, v_cache, cross_k, cross_v,
                out: torch.Tensor, pre=None, cross_rows: Optional[int] = None,
                attn_history: Optional[torch.Tensor] = None) -> "torch.cuda.CUDAGraph":
        """Capture one forward (fixed stage shape, slot table and buffers) into a hipGraph.  The forward is a pure launch
        sequence -- no host sync, no allocation -- so replaying it costs one graph launch instead of ~13 launches per
        layer.  `x`, `t`, `out` and the caches are captured BY ADDRESS: update their contents in place between replays
        (that is what the denoise loop does: latents and the timestep change, the shapes never do).
        `pre`: launches recorded in front of the forward (the i2v model type refreshes the latent channels of its 36-channel
        input buffer there)."""
        self.workspace(x.shape[0])                      # allocate outside the capture
        if x.shape[0] not in self._warm:                # one eager call per (engine, stage shape): the kernels a shape selects (pipelined
            # LayerNorm from 16 384 rows, the v8 GEMM variants, attn_cross_kernel<NT>, the split-KV tail) set their dynamic-LDS
            # attribute / query their occupancy on first launch -- outside the capture, not inside it
            if pre is not None:
                pre()
            self.forward(x, t, frame_ids, write_slots, visible_slots, k_cache, v_cache, cross_k, cross_v, out=out, cross_rows=cross_rows,
                         attn_history=attn_history)
            self._warm.add(x.shape[0])
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            if pre is not None:
                pre()
            self.forward(x, t, frame_ids, write_slots, visible_slots, k_cache, v_cache, cross_k, cross_v, out=out, cross_rows=cross_rows,
                         attn_history=attn_history)
        return g

