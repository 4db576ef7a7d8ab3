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


def handoff_to_initial_latent(vae, recv: torch.Tensor, return_first_frame: bool = False):
    """recv: [1, n, 16, h, w] = cat([frame0, anchors...]) (T2V n=8) or cat([frame0, f19, f20]) (I2V n=3).
    -> initial_latent [1, 2, 16, h, w] bf16 for the next chunk (with return_first_frame also the next chunk's first pixel
    frame [3, H, W] in [-1, 1]: the image the Wan-I2V model type is conditioned on from chunk 2 on)."""
    r = recv.to(torch.bfloat16)
    # mask latents [f0, f19, f19, f20, 0 ...]: only the first 4 matter for pixel frames 8..12 (causal decoder)
    lat = torch.stack([r[0, 0], r[0, -2], r[0, -2], r[0, -1]], dim=0).unsqueeze(0)
    px = vae.decode_to_pixel(lat).to(torch.bfloat16)                         # [1, 13, 3, H, W] in [-1, 1]
    px = (px * 0.5 + 0.5).clamp(0, 1).to(torch.bfloat16)
    clip = px[:, 8:13] * 2.0 - 1.0                                            # 5 frames -> first 2 latents (causal encoder)
    z = vae.encode_to_latent(clip.permute(0, 2, 1, 3, 4))
    if return_first_frame:
        return z[:, :2].to(torch.bfloat16), clip[0, 0]
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


def wavefront_layout(world: int, cfg_split: bool):
    """rank -> (lane, role) and the rank lists of every group, for `world` ranks of one node.

    cfg_split=False: `world` chunk lanes, one rank each (chunk c on lane c % world).
    cfg_split=True : world/2 lanes of two ranks (2l, 2l+1) = (cond, uncond) -- the reference's device_cond/device_uncond
    seam (casual_fps_inference.py:42-43).  The wavefront is only ~3.1 chunks deep, so 8 GPUs are used
    as 4 lanes x 2 rather than 8 lanes.  Returns dict(lanes=[[ranks]], lane_of=[...], role_of=[...], heads=[...])."""
    if cfg_split:
        if world % 2:
            raise ValueError("cfg_split needs an even number of ranks")
        lanes = [[2 * i, 2 * i + 1] for i in range(world // 2)]
    else:
        lanes = [[r] for r in range(world)]
    lane_of = [0] * world
    role_of = [0] * world
    for li, ranks in enumerate(lanes):
        for role, r in enumerate(ranks):
            lane_of[r], role_of[r] = li, role
    return dict(lanes=lanes, lane_of=lane_of, role_of=role_of, heads=[ranks[0] for ranks in lanes])


class CfgPair:
    """The two ranks that share one chunk: role 0 runs the conditional branch, role 1 the unconditional one.

    exchange(): one all-gather of the two flow predictions ([nF,16,h,w] bf16 each, <= 3.2 MB at 720p) per denoise step
    over the pair's xGMI link; broadcast(): role 0's tensor to role 1 (noise, re-noise draws, initial latents)."""

    def __init__(self, ranks: Sequence[int], group, device):
        import torch.distributed as dist
        self.dist, self.group, self.ranks = dist, group, list(ranks)
        self.role = self.ranks.index(dist.get_rank())
        self.backend = dist.get_backend(group)
        self.device = torch.device("cpu") if self.backend == "gloo" else torch.device(device)

    @classmethod
    def build(cls, world: int, device, cfg_split: bool = True, warm: bool = True):
        """Collective over ALL ranks (every rank creates every group, in the same order, as torch.distributed requires -- with the
        nccl backend and `init_process_group(device_id=...)` a sub-group is an `ncclCommSplit` of the world communicator, itself a
        collective over the parent).  Returns (pair or None, lane_heads_group, layout).
        warm: every member then runs one tiny all-reduce per group IN CREATION ORDER (its pair group, then -- lane heads only -- the
        heads group): an RCCL communicator that was not created eagerly is created by its first collective, a blocking rendezvous of
        all its members; doing it here, in one fixed order on every rank, keeps it out of the data path, where the lane heads' first
        use of the heads group (an anchor hand-off on the side stream) and of the pair group (a flow all-gather on the compute stream)
        would otherwise come in a data-dependent order."""
        import torch.distributed as dist
        lay = wavefront_layout(world, cfg_split)
        me = dist.get_rank()
        pair = None
        if cfg_split:
            for ranks in lay["lanes"]:
                g = dist.new_group(ranks)
                if me in ranks:
                    pair = cls(ranks, g, device)
        heads = dist.new_group(lay["heads"]) if cfg_split else None
        if warm and cfg_split:
            one = torch.ones(1, device=pair.device)
            dist.all_reduce(one, group=pair.group)
            if me in lay["heads"]:
                dist.all_reduce(one, group=heads)
            assert float(one) == (2.0 * len(lay["heads"]) if me in lay["heads"] else 2.0), float(one)
        return pair, heads, lay

    def exchange(self, mine: torch.Tensor, both: torch.Tensor) -> torch.Tensor:
        """both[role] <- the flow prediction of that role, on both ranks.  `both`: [2, *mine.shape]."""
        assert both.shape[0] == 2 and both.shape[1:] == mine.shape
        if self.backend == "gloo":
            m = mine.cpu() if mine.device.type != "cpu" else mine
            parts = [torch.empty_like(m), torch.empty_like(m)]
            self.dist.all_gather(parts, m.contiguous(), group=self.group)
            both[0].copy_(parts[0])
            both[1].copy_(parts[1])
        else:
            self.dist.all_gather_into_tensor(both, mine, group=self.group)
        return both

    def broadcast(self, t: torch.Tensor) -> torch.Tensor:
        if self.backend == "gloo" and t.device.type != "cpu":
            c = t.cpu()
            self.dist.broadcast(c, src=self.ranks[0], group=self.group)
            return t.copy_(c)
        self.dist.broadcast(t, src=self.ranks[0], group=self.group)
        return t


class ChunkHandoff:
    """Point-to-point anchor exchange between the ranks of one node (or of `group`, e.g. the lane heads of a CFG-split
    layout; ranks inside the group are translated to global ranks for the p2p calls).

    Ready handshake (round 4).  An RCCL send is a KERNEL that spins on the sender's GPU until the peer's matching recv kernel
    runs.  In the wavefront the consumer of chunk c + 1 is usually already waiting (lanes >= the ~3.1 chunks the wavefront keeps
    busy), but on wrap-around with few lanes (n_chunks > lanes: BASELINE configs[4], or 2 GPUs) it is still computing its
    previous chunk when the producer's anchor stage ends -- the send kernel would then sit on a CU for minutes next to kernels
    that assume one workgroup per CU (attn_w64_kernel: 128 KiB of LDS and all 512 registers of every SIMD; the GEMM's tile
    tickets).  So the DATA goes over RCCL, but WHEN it is issued is agreed on the host: the consumer announces "ready for
    chunk c" (an 8-byte gloo message on a side group) right before it posts its recv; the producer issues the send only
    once that announcement has arrived -- at the sink if it already has, else at the next poll point (the pipeline calls
    `poll()` between stages, `run_chunk_wavefront` drains with a blocking wait after the chunk).  Either way a send kernel is
    only ever launched against a recv that is already posted: it occupies one CU for the microseconds the 3.7 MB take.
    Chunk order makes the blocking drain deadlock-free: the wait is for the consumer to finish an EARLIER chunk.
    `stats[c]` keeps wall-clock stamps (time.time(), one host) of every hand-off for bench.py's wavefront report.

    `control_timeout`: timeout of the gloo control group (default 12 h, the data path's own bound in mmpl_amd/cli.py): the
    producer posts the receive of "ready for chunk c + 1" when ITS chunk starts and a waiter thread waits on it at once, so with
    torch's 30-minute gloo default a consumer more than 30 minutes behind would abort a producer whose data path would have waited.
    When the data group itself is gloo it is also the control group and carries whatever timeout its creator gave it.

    `loopback` (single-rank self-test of the transport, tests/test_rccl_loopback_gpu.py): a hand-off whose consumer is this very
    rank normally never touches the transport (`_local`); with loopback it goes through it -- a grouped send + recv to our own rank
    on the side stream (RCCL runs that as a copy kernel), behind the same deferred-issue bookkeeping, with the gloo control group
    created next to the RCCL group.  gloo has no pair to oneself, so the "ready" announcement is delivered in-process."""

    def __init__(self, shape: Sequence[int], device, group=None, ready_handshake: bool = True, control_timeout=None,
                 loopback: bool = False):
        import datetime
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device("cpu") if self.backend == "gloo" else torch.device(device)
        self.shape = tuple(shape)
        self._g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._pending: List = []
        self._deferred: List = []              # [(chunk, hdr, payload, ready request)] hand-offs whose consumer is not ready yet
        self._ready_req = {}                   # consumer chunk -> irecv request of its "ready" announcement
        self._keep: List = []                  # control tensors / requests that must outlive their call
        self.stats = {}
        self.loopback = bool(loopback) and self.world == 1
        if self.loopback and self.backend == "gloo":
            raise ValueError("ChunkHandoff(loopback=True) needs a transport that can send to its own rank (RCCL); gloo has no pair to oneself")
        self.handshake = ready_handshake and (self.world > 1 or self.loopback)
        self._ctl = None
        self._self_ready = {}                  # loopback: chunk -> threading.Event of our own announcement
        if self.handshake:
            # control plane: gloo.  The data group itself when it is gloo (CPU tests, shared-GPU functional runs); else a gloo
            # group over the same ranks, created by its members only (use_local_synchronization: non-members do not take part)
            if self.backend == "gloo":
                self._ctl = group
            else:
                ranks = [self._g(r) for r in range(self.world)]
                self._ctl = dist.new_group(ranks, backend="gloo", use_local_synchronization=True,
                                           timeout=control_timeout or datetime.timedelta(hours=12))

    def owner(self, chunk: int) -> int:
        return chunk % self.world

    # ---- ready handshake (host side, gloo)
    def expect_ready(self, chunk: int) -> None:
        """Producer of `chunk - 1`: post the receive of the consumer's "ready for `chunk`" announcement (returns at once)."""
        if not self.handshake or chunk in self._ready_req:
            return
        import threading
        if self.owner(chunk) == self.rank:
            if self.loopback:                                  # our own announcement: an in-process Event (gloo has no self pair)
                self._ready_req[chunk] = (None, None, self._self_ready.setdefault(chunk, threading.Event()), [])
            return
        buf = torch.zeros(1, dtype=torch.int64)
        req = self.dist.irecv(buf, self._g(self.owner(chunk)), group=self._ctl, tag=_READY_TAG + chunk)
        # A gloo receive's Work.is_completed() stays False until somebody WAITS on it, however long ago the message landed
        # (measured: tools-free probe in profiles/NOTEBOOK_r04.md section J) -- polling it would defer every hand-off to the blocking
        # drain at the end of the producer's chunk and serialise the wavefront.  So a daemon thread does the waiting (Work.wait()
        # releases the GIL) and an Event carries the news to the non-blocking checks at the sink and at the stage boundaries.
        arrived, failed = threading.Event(), []

        def waiter():
            try:
                req.wait()
            except Exception as e:                             # timeout / peer gone: surfaces at the next readiness check
                failed.append(e)
            finally:
                arrived.set()
        threading.Thread(target=waiter, name=f"mmpl-ready-{chunk}", daemon=True).start()
        self._ready_req[chunk] = (req, buf, arrived, failed)

    def announce_ready(self, chunk: int) -> None:
        """Consumer of `chunk`: tell the producer of chunk - 1 that the recv is about to be posted."""
        import time
        self.stats.setdefault(chunk, {})["t_ready"] = time.time()
        if self.handshake and self.loopback and self.owner(chunk - 1) == self.rank:
            import threading
            self._self_ready.setdefault(chunk, threading.Event()).set()
            return
        if not self.handshake or self.owner(chunk - 1) == self.rank:
            return
        buf = torch.tensor([chunk], dtype=torch.int64)
        self._keep.append((self.dist.isend(buf, self._g(self.owner(chunk - 1)), group=self._ctl, tag=_READY_TAG + chunk), buf))

    def _consumer_ready(self, chunk: int, block: bool, timeout: Optional[float] = None) -> bool:
        if not self.handshake:
            return True
        if chunk not in self._ready_req:
            self.expect_ready(chunk)
        _, _, arrived, failed = self._ready_req[chunk]
        if block:
            arrived.wait(timeout)
        if arrived.is_set() and failed:
            raise RuntimeError(f"hand-off: waiting for the consumer of chunk {chunk} to announce itself failed: {failed[0]}")
        return arrived.is_set()

    def _issue(self, chunk: int, hdr: torch.Tensor, payload: torch.Tensor) -> None:
        import time
        dst = self._g(self.owner(chunk + 1))
        self.stats.setdefault(chunk + 1, {})["t_issued"] = time.time()
        if self.loopback and self.owner(chunk + 1) == self.rank:
            # to ourselves through the transport: sends and receives in ONE group call (an ungrouped send to oneself never meets its
            # receive); the received copies are what recv() hands out
            rh, rp = torch.zeros_like(hdr), torch.empty_like(payload)
            P = self.dist.P2POp
            ops = [P(self.dist.isend, hdr, dst, self.group), P(self.dist.irecv, rh, dst, self.group),
                   P(self.dist.isend, payload, dst, self.group), P(self.dist.irecv, rp, dst, self.group)]
            if self._side is not None:
                self._side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(self._side):
                    self._pending += self.dist.batch_isend_irecv(ops)
                    for t in (hdr, payload, rh, rp):
                        t.record_stream(self._side)
            else:
                self._pending += self.dist.batch_isend_irecv(ops)
            self._keep.append((hdr, payload))
            self._local = (rh, rp)
            return
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                self._pending += [self.dist.isend(hdr, dst, group=self.group, tag=2 * chunk),
                                  self.dist.isend(payload, dst, group=self.group, tag=2 * chunk + 1)]
                payload.record_stream(self._side)
                hdr.record_stream(self._side)
        else:
            self._pending += [self.dist.isend(hdr, dst, group=self.group, tag=2 * chunk),
                              self.dist.isend(payload, dst, group=self.group, tag=2 * chunk + 1)]
        self._keep.append((hdr, payload))

    def send(self, chunk: int, tensor: Optional[torch.Tensor], status: int = OK) -> None:
        """Hand chunk `chunk`'s anchors to the owner of chunk+1 (asynchronous; overlaps the caller's next stage).  Issued now if
        the consumer has announced it is ready (or the handshake is off), otherwise at the next poll() / drain()."""
        import time
        dst_local = self.owner(chunk + 1)
        now = time.time()
        self.stats.setdefault(chunk + 1, {})["t_sink"] = now
        hdr = torch.tensor([status, chunk, int(now * 1e6)], dtype=torch.int64, device=self.device)
        payload = (torch.zeros(self.shape, dtype=torch.bfloat16, device=self.device) if tensor is None
                   else tensor.detach().to(device=self.device, dtype=torch.bfloat16).reshape(self.shape).contiguous())
        if dst_local == self.rank and not self.loopback:       # world size 1 (or wrap onto ourselves): local hand-over
            self._local = (hdr, payload)
            return
        if payload.data_ptr() == (tensor.data_ptr() if tensor is not None else 0):
            payload = payload.clone()                          # a deferred send must not alias a buffer the caller keeps writing
        if self._consumer_ready(chunk + 1, block=False):
            self._issue(chunk, hdr, payload)
        else:
            self._deferred.append((chunk, hdr, payload))

    def poll(self) -> int:
        """Issue the deferred hand-offs whose consumers have become ready (non-blocking); returns how many are still deferred."""
        still = []
        for chunk, hdr, payload in self._deferred:
            if self._consumer_ready(chunk + 1, block=False):
                self._issue(chunk, hdr, payload)
            else:
                still.append((chunk, hdr, payload))
        self._deferred = still
        return len(still)

    def drain(self, timeout: Optional[float] = None) -> None:
        """Blocking: wait for every deferred hand-off's consumer and issue it (end of the producer's chunk).  A hand-off to
        ourselves (loopback) stays deferred: its consumer is this thread, recv() issues it.  `timeout` (seconds per hand-off):
        give up on a consumer that does not announce itself (error paths only) -- the entry is dropped, the peer's recv then
        runs into the process group's own timeout."""
        still = []
        for chunk, hdr, payload in self._deferred:
            if self.loopback and self.owner(chunk + 1) == self.rank:
                still.append((chunk, hdr, payload))
                continue
            if self._consumer_ready(chunk + 1, block=True, timeout=timeout):
                self._issue(chunk, hdr, payload)
        self._deferred = still

    def fail_deferred(self, timeout: float = 600.0) -> None:
        """The producer is about to raise with hand-offs still deferred (its consumer was busy at the sink and at every poll since):
        they would be lost and the consumer would sit in recv until the process-group timeout.  Their status becomes FAILED -- the
        video cannot be completed, the consumer must raise, not go on -- and they are issued as soon as the consumer announces
        itself, waiting at most `timeout` seconds for that."""
        for _, hdr, _ in self._deferred:
            hdr[0] = FAILED
        self.drain(timeout=timeout)

    def recv(self, chunk: int) -> torch.Tensor:
        """Receive the hand-off produced by chunk-1.  Blocks until the producer's anchor stage is done (minutes per lane at
        14B/720p), bounded only by the process group's timeout -- the entry points raise it from torch's 10-minute default
        to 12 h (mmpl_amd/cli.py) / 4 h (bench.py); raises if the producer reported failure."""
        import time
        src_local = self.owner(chunk - 1)
        src = self._g(src_local)
        self.announce_ready(chunk)
        if src_local == self.rank:
            if self.loopback:                                   # our own deferred hand-off, through the transport, now that "we" are ready
                self.poll()
                for w in self._pending:
                    w.wait()
                self._pending.clear()
                if self._side is not None:
                    torch.cuda.current_stream(self.device).wait_stream(self._side)
            hdr, payload = self._local
        else:
            hdr = torch.zeros(3, dtype=torch.int64, device=self.device)
            payload = torch.empty(self.shape, dtype=torch.bfloat16, device=self.device)
            self.dist.recv(hdr, src, group=self.group, tag=2 * (chunk - 1))
            self.dist.recv(payload, src, group=self.group, tag=2 * (chunk - 1) + 1)
        st, ck, t_sink_us = [int(v) for v in hdr.tolist()]      # (.tolist() synchronises: the payload has landed too -- same stream)
        rec = self.stats.setdefault(chunk, {})
        rec["t_recv_done"] = time.time()
        rec["t_sink"] = t_sink_us * 1e-6
        if st != OK or ck != chunk - 1:
            raise RuntimeError(f"hand-off for chunk {chunk}: producer rank {src} reported status {st} (chunk {ck})")
        return payload

    def flush(self) -> None:
        self.drain()
        for w in self._pending:
            w.wait()
        self._pending.clear()
        for item in self._keep:
            if hasattr(item[0], "wait"):
                item[0].wait()
        self._keep.clear()
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)


_READY_TAG = 1 << 20


def run_chunk_wavefront(make_chunk: Callable[[int, Optional[torch.Tensor], Callable[[torch.Tensor], None]], torch.Tensor],
                        n_chunks: int, handoff: Optional[ChunkHandoff], to_initial: Callable[[torch.Tensor], torch.Tensor],
                        gather: bool = True, pair: Optional[CfgPair] = None, lane: int = 0, n_lanes: int = 1,
                        initial_like: Optional[torch.Tensor] = None) -> Optional[List[torch.Tensor]]:
    """Chunk c runs on rank c % W (the reference's pipeline k <-> cuda:k and its round-robin,
    Wan_fps_inference_parallel_4gpu_5-60s.py:252-332).

    make_chunk(c, initial_latent_or_None, sink) must call sink(handoff_tensor) once the anchor stage is done and return
    the chunk's result tensor.  Returns the list of all chunk results on rank 0 (None elsewhere) when gather=True.

    CFG-split lanes: the lane heads (pair.role == 0) run exactly the loop above over `handoff` (built on the heads
    group); their uncond partners (pair.role == 1, handoff=None) run the same chunks of lane `lane` of `n_lanes` with a
    placeholder `initial_like` that the pipeline overwrites with the head's broadcast, and take no part in the gather."""
    if pair is not None and pair.role != 0:
        for c in range(lane, n_chunks, n_lanes):
            make_chunk(c, None if c == 0 else torch.empty_like(initial_like), lambda t: None)
        return None
    dist = handoff.dist
    mine = {}
    for c in range(handoff.rank, n_chunks, handoff.world):
        sent = [False]

        def sink(t, c=c):
            if c + 1 < n_chunks:
                handoff.send(c, t)
            sent[0] = True

        try:
            if c + 1 < n_chunks:
                handoff.expect_ready(c + 1)                    # the consumer's announcement may arrive any time from now on
            initial = to_initial(handoff.recv(c)) if c > 0 else None
            mine[c] = make_chunk(c, initial, sink)
            if not sent[0] and c + 1 < n_chunks:
                raise RuntimeError(f"chunk {c} finished without producing its hand-off")
            handoff.drain()                                    # a consumer that was still busy at the sink and at every poll since
        except Exception:
            if c + 1 < n_chunks:
                try:
                    if not sent[0]:
                        handoff.send(c, None, FAILED)         # unblock the consumer with an error instead of a hang
                    # ... and a hand-off the ready handshake deferred at the sink (sent[0] is True, nothing has left yet) must not
                    # be lost either: it goes out marked FAILED as soon as the consumer announces itself
                    handoff.fail_deferred()
                    handoff.flush()
                except Exception:                              # (the consumer may be gone as well: keep the ORIGINAL error)
                    pass
            raise
    handoff.flush()
    if not gather:
        return None
    return gather_chunks(mine, n_chunks, handoff)


def gather_chunks(mine: dict, n_chunks: int, handoff: "ChunkHandoff") -> Optional[List[torch.Tensor]]:
    """All-gather of the chunk results over the hand-off group, on the device (RCCL ``all_gather_into_tensor`` over xGMI; the
    in-process list + ``torch.cat(video_chunks)`` of fastapi_parallel_i2v_server.py:851-856): round r collects chunks
    r*W .. r*W + W - 1, one per rank, into ONE [W, ...] device buffer -- nothing is pickled and nothing passes through the
    host (a decoded 720p chunk is 224 MB as uint8 frames, 896 MB as fp32).  Every chunk result must have the same shape and
    dtype (ranks without a chunk in the last round contribute a dummy).  Rank 0 of the group gets the list ordered by chunk
    index (tensors on the hand-off's device: the GPU for RCCL, the host for gloo), the other ranks None."""
    dist, W, me = handoff.dist, handoff.world, handoff.rank
    meta = torch.zeros(10, dtype=torch.int64, device=handoff.device)        # [has, dtype code, ndim, shape...] from a rank that owns a chunk
    if mine:
        t0 = next(iter(mine.values()))
        assert t0.dim() <= 7
        meta[:3 + t0.dim()] = torch.tensor([1, _DTYPES.index(t0.dtype), t0.dim(), *t0.shape], dtype=torch.int64)
    metas = [torch.zeros_like(meta) for _ in range(W)]
    dist.all_gather(metas, meta, group=handoff.group)
    ref = next((m for m in metas if int(m[0])), None)
    if ref is None:
        return [] if me == 0 else None
    ref = [int(v) for v in ref.tolist()]
    dtype, shape = _DTYPES[ref[1]], tuple(ref[3:3 + ref[2]])
    out: List[Optional[torch.Tensor]] = [None] * n_chunks
    for r in range((n_chunks + W - 1) // W):
        c = r * W + me
        part = mine.get(c) if c < n_chunks else None
        if part is None:
            part = torch.zeros(shape, dtype=dtype, device=handoff.device)
        else:
            assert tuple(part.shape) == shape and part.dtype == dtype, "every chunk result must have the same shape and dtype"
            part = part.detach().to(handoff.device).contiguous()
        if handoff.backend == "gloo":
            parts = [torch.empty(shape, dtype=dtype) for _ in range(W)]
            dist.all_gather(parts, part, group=handoff.group)
        else:
            both = torch.empty((W,) + shape, dtype=dtype, device=handoff.device)
            dist.all_gather_into_tensor(both, part, group=handoff.group)
            parts = list(both.unbind(0))
        if me == 0:
            for j in range(W):
                if r * W + j < n_chunks:
                    out[r * W + j] = parts[j]
    return out if me == 0 else None


_DTYPES = [torch.float32, torch.bfloat16, torch.float16, torch.uint8, torch.int64, torch.int32, torch.float64]
