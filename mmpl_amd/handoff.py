"""Inter-chunk hand-off and the chunk wavefront across ranks (one process per GPU, torch.distributed = RCCL/xGMI).

Replaces the reference's only inter-GPU data flow: ``torch.save(save_latents, "latents_chunk{k}.pt")`` in one thread and
a 1 Hz ``os.path.exists`` poll + ``torch.load`` + ``os.remove`` in another
(MMPL_t2v/pipeline/casual_fps_inference.py:380-383, MMPL_t2v/Wan_fps_inference_parallel_4gpu_20s.py:184-188) with a
point-to-point send/recv of the 1.6-3.7 MB anchor tensor rank c%W -> (c+1)%W, issued on a side stream right after the
anchor stage so it overlaps the sender's in-fill stages, plus a status header so that a failed producer makes its
consumer raise instead of spinning forever ("failure detection").

Also the consumer-side transform (parallel_4gpu_20s.py:191-205) and the single-GPU rolling variant
(Wan_fps_inference_1gpu.py:177-186), both computed on the causal *prefix* only: decoding 4 latents / encoding 5 pixel
frames gives bit-identical results to the reference's 21-latent decode + 81-frame encode (probed).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

OK, FAILED = 1, -1


def handoff_to_initial_latent(vae, recv: torch.Tensor) -> torch.Tensor:
    """recv: [1, n, 16, h, w] = cat([frame0, anchors...]) (T2V n=8) or cat([frame0, f19, f20]) (I2V n=3).
    -> initial_latent [1, 2, 16, h, w] bf16 for the next chunk."""
    r = recv.to(torch.bfloat16)
    # mask latents [f0, f19, f19, f20, 0 ...]: only the first 4 matter for pixel frames 8..12 (causal decoder)
    lat = torch.stack([r[0, 0], r[0, -2], r[0, -2], r[0, -1]], dim=0).unsqueeze(0)
    px = vae.decode_to_pixel(lat).to(torch.bfloat16)                         # [1, 13, 3, H, W] in [-1, 1]
    px = (px * 0.5 + 0.5).clamp(0, 1).to(torch.bfloat16)
    clip = px[:, 8:13] * 2.0 - 1.0                                            # 5 frames -> first 2 latents (causal encoder)
    z = vae.encode_to_latent(clip.permute(0, 2, 1, 3, 4))
    return z[:, :2].to(torch.bfloat16)


def rolling_initial_latent(vae, video: torch.Tensor) -> torch.Tensor:
    """Single-GPU rollout (Wan_fps_inference_1gpu.py:177-186): last 5 pixel frames of a finished chunk ([1, T, 3, H, W] in
    [0, 1]) -> the next chunk's 2 initial latents."""
    clip = video[:, -5:].to(torch.bfloat16) * 2.0 - 1.0
    return vae.encode_to_latent(clip.permute(0, 2, 1, 3, 4))[:, :2].to(torch.bfloat16)


def stitch_chunks(videos: Sequence[torch.Tensor], num_overlap_frames: int = 2) -> torch.Tensor:
    """Wan_fps_inference_1gpu.py:192-198 / fastapi_parallel_i2v_server.py:851-856: drop the (n-1)*4+1 = 5 overlapping pixel
    frames of every chunk after the first and concatenate along time.  videos: [B, T, ...] each."""
    drop = (num_overlap_frames - 1) * 4 + 1
    return torch.cat([v if i == 0 else v[:, drop:] for i, v in enumerate(videos)], dim=1)


class ChunkHandoff:
    """Point-to-point anchor exchange between the ranks of one node."""

    def __init__(self, shape: Sequence[int], device, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device("cpu") if self.backend == "gloo" else torch.device(device)
        self.shape = tuple(shape)
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._pending: List = []

    def owner(self, chunk: int) -> int:
        return chunk % self.world

    def send(self, chunk: int, tensor: Optional[torch.Tensor], status: int = OK) -> None:
        """Send chunk `chunk`'s hand-off to the owner of chunk+1 (asynchronous; overlaps the caller's next stage)."""
        dst = self.owner(chunk + 1)
        hdr = torch.tensor([status, chunk], dtype=torch.int64, device=self.device)
        payload = (torch.zeros(self.shape, dtype=torch.bfloat16, device=self.device) if tensor is None
                   else tensor.detach().to(device=self.device, dtype=torch.bfloat16).reshape(self.shape).contiguous())
        if dst == self.rank:                                   # world size 1 (or wrap onto ourselves): local hand-over
            self._local = (hdr, payload)
            return
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                self._pending += [self.dist.isend(hdr, dst, group=self.group, tag=2 * chunk),
                                  self.dist.isend(payload, dst, group=self.group, tag=2 * chunk + 1)]
                payload.record_stream(self._side)
        else:
            self._pending += [self.dist.isend(hdr, dst, group=self.group, tag=2 * chunk),
                              self.dist.isend(payload, dst, group=self.group, tag=2 * chunk + 1)]

    def recv(self, chunk: int) -> torch.Tensor:
        """Receive the hand-off produced by chunk-1.  Bounded by the process group's timeout; raises if the producer
        reported failure."""
        src = self.owner(chunk - 1)
        if src == self.rank:
            hdr, payload = self._local
        else:
            hdr = torch.zeros(2, dtype=torch.int64, device=self.device)
            payload = torch.empty(self.shape, dtype=torch.bfloat16, device=self.device)
            self.dist.recv(hdr, src, group=self.group, tag=2 * (chunk - 1))
            self.dist.recv(payload, src, group=self.group, tag=2 * (chunk - 1) + 1)
        st, ck = [int(v) for v in hdr.tolist()]
        if st != OK or ck != chunk - 1:
            raise RuntimeError(f"hand-off for chunk {chunk}: producer rank {src} reported status {st} (chunk {ck})")
        return payload

    def flush(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)


def run_chunk_wavefront(make_chunk: Callable[[int, Optional[torch.Tensor], Callable[[torch.Tensor], None]], torch.Tensor],
                        n_chunks: int, handoff: ChunkHandoff, to_initial: Callable[[torch.Tensor], torch.Tensor],
                        gather: bool = True) -> Optional[List[torch.Tensor]]:
    """Chunk c runs on rank c % W (the reference's pipeline k <-> cuda:k and its round-robin,
    Wan_fps_inference_parallel_4gpu_5-60s.py:252-332).

    make_chunk(c, initial_latent_or_None, sink) must call sink(handoff_tensor) once the anchor stage is done and return
    the chunk's result tensor.  Returns the list of all chunk results on rank 0 (None elsewhere) when gather=True."""
    dist = handoff.dist
    mine = {}
    for c in range(handoff.rank, n_chunks, handoff.world):
        sent = [False]

        def sink(t, c=c):
            if c + 1 < n_chunks:
                handoff.send(c, t)
            sent[0] = True

        try:
            initial = to_initial(handoff.recv(c)) if c > 0 else None
            mine[c] = make_chunk(c, initial, sink)
            if not sent[0] and c + 1 < n_chunks:
                raise RuntimeError(f"chunk {c} finished without producing its hand-off")
        except Exception:
            if not sent[0] and c + 1 < n_chunks:
                handoff.send(c, None, FAILED)                 # unblock the consumer with an error instead of a hang
                handoff.flush()
            raise
    handoff.flush()
    if not gather:
        return None
    objs = [None] * handoff.world if handoff.rank == 0 else None
    dist.gather_object({c: v.cpu() for c, v in mine.items()}, objs, dst=0, group=handoff.group)
    if handoff.rank != 0:
        return None
    merged = {}
    for o in objs:
        merged.update(o)
    return [merged[c] for c in range(n_chunks)]
