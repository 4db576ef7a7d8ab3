"""``CausalFPSInferencePipeline`` -- drop-in for MMPL_t2v/pipeline/casual_fps_inference.py (and the I2V variant,
MMPL_i2v/pipeline/casual_fps_inference.py) with the denoising hot loop on the HIP engines.

Same constructor / ``inference()`` signature, attributes (``generator_cond``, ``vae``, ``text_encoder``,
``independent_first_frame``, ``need_wait``, ``save``) and stage semantics.  What changed
underneath, MI355X-first:
  * one DiT forward = one C call (``mmpl_dit_forward``), K/V read in place through the slot table -- no gather copies,
    no host syncs inside a forward;
  * CFG + UniPC = one fused kernel per step (``mmpl_cfg_unipc_step``); with hipGraphs on, a whole denoise step (both
    forwards + that kernel, its scalars and the next timestep read from device tables) is ONE graph replayed 50 times;
  * nothing is shuffled to the CPU (T5 / VAE stay resident: 288 GB HBM);
  * the literals 1560 / 40x128 / 40 blocks are derived from a ``Geometry`` and the model config;
  * the hand-off is delivered to ``self.handoff_sink`` (default: ``torch.save(self.save)`` like the reference; the
    multi-GPU runner installs an RCCL send, mmpl_amd/handoff.py);
  * the reference's ``device_cond`` / ``device_uncond`` seam (two GPUs per pipeline, :42-43,346-367) is ``self.cfg_pair``
    in the one-process-per-GPU world: the two ranks of a ``CfgPair`` each own ONE branch (cond or uncond: its KV cache,
    cross-attention cache and forward), exchange the two flow predictions per step (one 2-rank all-gather of <= 3.2 MB)
    and both apply the fused CFG + UniPC update, so their latents stay bit-identical without a second exchange.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from ..geometry import Geometry
from ..scheduler import FlowUniPCMultistepScheduler
from ..stage_plan import StagePlan
from ..wan_wrapper import WanFPSWrapper, WanTextEncoder, WanVAEWrapper


class CausalFPSInferencePipeline(torch.nn.Module):
    def __init__(self, args, device, generator=None, text_encoder=None, vae=None, device_cond="cuda:0",
                 device_uncond="cuda:0", save="latents_chunk1.pt", mode: str = "t2v", geometry: Optional[Geometry] = None):
        super().__init__()
        self.need_wait = False
        self.save = save
        self.device_cond = device_cond
        self.device_uncond = device_uncond
        self.mode = mode
        self.plan = StagePlan(mode)
        self.geometry = geometry or (generator.geometry if generator is not None else Geometry.named("480p"))

        self.generator_cond = WanFPSWrapper(**getattr(args, "model_kwargs", {}), is_causal=True, geometry=self.geometry,
                                            device=device_cond) if generator is None else generator
        self.generator_cond.model.num_frame_per_block = 1
        self.text_encoder = WanTextEncoder(device=device_cond) if text_encoder is None else text_encoder
        self.vae = WanVAEWrapper(geometry=self.geometry, device=device_cond) if vae is None else vae

        self.num_train_timesteps = args.num_train_timestep
        self.sampling_steps = getattr(args, "sampling_steps", 50)
        self.sample_solver = "unipc"
        self.shift = args.timestep_shift
        self.num_transformer_blocks = self.generator_cond.engine.L
        self.frame_seq_length = self.geometry.frame_seqlen

        self.kv_cache_pos = None
        self.kv_cache_neg = None
        self.crossattn_cache_pos = None
        self.crossattn_cache_neg = None
        self.args = args
        self.num_frame_per_block = 1
        self.independent_first_frame = args.independent_first_frame
        self.local_attn_size = -1
        self.handoff_sink: Optional[Callable[[torch.Tensor], None]] = None
        # called at every stage boundary after the hand-off stage (the host is in step with the GPU there): the multi-GPU runner
        # installs ChunkHandoff.poll, which issues a deferred RCCL send once its consumer has announced it is ready (handoff.py)
        self.handoff_poll: Optional[Callable[[], object]] = None
        self.renoise_override = None      # tests: {frame: [1,16,h,w]} instead of torch.randn_like draws
        self.use_graphs = True            # one hipGraph per (stage, cond|uncond) forward, replayed 50 + 1 times
        self.step_graphs = True           # ... and one per whole denoise step (2 forwards + CFG/UniPC) for the 50 steps
        # the cond and the uncond forward of a step as two PARALLEL branches of the step graph (second stream, private workspace):
        # None = where it pays (mmpl_amd.stage_plan.concurrent_cfg_pays: small stages / small models), True / False = always / never.
        # Same kernels, same arguments, same bits either way (tests/test_pipeline_gpu.py).
        self.concurrent_cfg: Optional[bool] = None
        self._side_stream = None
        # Block 0's self-attention is the same computation in both CFG branches (same latents, same timestep, the same layer-0 K / V
        # in both caches: every forward this pipeline issues runs on both branches); in the sequential step graphs the uncond forward
        # takes x after that residual from the cond forward instead of recomputing it (mmpl_dit_forward share_out / share_in;
        # bit-identical).  False switches it off.
        self.share_block0 = True
        # The self-attention kernel's softmax is a max-free FAST pass per 256-row query block plus a GENERAL pass for the blocks it
        # cannot hold (heavy-tailed scores: large QK-norm gains).  With a history (a state byte + 128 lane references per layer / head /
        # block, kept with each branch's KV cache, zeroed per stage) a block that failed takes its next FAST reference from what the last
        # pass learned -- and, failing that too, starts in GENERAL -- instead of paying for both passes on every denoise step.  Exact
        # softmax either way; for such blocks the bits then depend on the steps before (include/mmpl_hip.h).  False = stateless.
        self.attn_history = True
        self.cfg_pair = None              # mmpl_amd.handoff.CfgPair: this rank runs only the cond (role 0) / uncond (role 1) branch

        # ---- "add new noise on previous frames" schedule (casual_fps_inference.py:93-108); the randint keeps the
        # reference's RNG consumption order; the resulting timestep is >= 1000, i.e. pure noise
        self.ddpm_scheduler = self.generator_cond.get_scheduler()
        self.image_or_video_shape = [1, 1, 16, self.geometry.lat_h, self.geometry.lat_w]
        self.ddpm_index = torch.randint(980, self.num_train_timesteps, [1, 1], device=self.device_cond, dtype=torch.long)
        self.ddmp_timestep = self.ddpm_scheduler.timesteps.to(self.ddpm_index.device)[self.ddpm_index] + 1000

    def to(self, *args, **kwargs):
        return self

    # ------------------------------------------------------------------------------------------------------------
    def _forward(self, latents, cond, timestep, kv, cross, frames, out=None, workspace=None, share_out=None, share_in=None):
        S = self.frame_seq_length
        starts = [f * S for f in frames]
        return self.generator_cond(noisy_image_or_video=latents, conditional_dict=cond, timestep=timestep, kv_cache=kv,
                                   crossattn_cache=cross, current_start=starts, cache_start=starts, out=out, workspace=workspace,
                                   share_out=share_out, share_in=share_in)[0]

    def _branches(self, cond, uncond):
        """[(conditional_dict, kv_cache, crossattn_cache, index into the flow pair)] this rank computes."""
        b = [(cond, self.kv_cache_pos, self.crossattn_cache_pos, 0), (uncond, self.kv_cache_neg, self.crossattn_cache_neg, 1)]
        return b if self.cfg_pair is None else [b[self.cfg_pair.role]]

    def _refresh(self, latents, cond, uncond, timestep, frames):
        """rerun with timestep zero to update the KV cache with clean context (:385-403)."""
        for d, kv, cross, _ in self._branches(cond, uncond):
            self._forward(latents, d, timestep * 0, kv, cross, frames)

    def inference(self, noise: torch.Tensor, text_prompts: List[str], initial_latent: Optional[torch.Tensor] = None,
                  return_latents: bool = False, start_frame_index: Optional[int] = 0, decode: bool = True,
                  image_condition: Optional[dict] = None):
        """noise: [1, 21, 16, h, w]; initial_latent: [1, n, 16, h, w] (T2V chunks >= 2: n = 2; I2V: n = 1 image latent,
        chunks >= 2: n = 2).  Returns video [1, T, 3, 8h, 8w] in [0, 1] (and the latents).
        image_condition: {"clip_fea": [257, 1280], "y": [20, 21, h, w]} (mmpl_amd.i2v_condition.build_image_condition) --
        required when the generator is the Wan-I2V model type; both CFG branches get the same one, as in
        wan/image2video.py:283-295 (arg_c / arg_null)."""
        batch_size, num_frames, num_channels, height, width = noise.shape
        assert batch_size == 1 and num_frames == self.geometry.frames_per_chunk
        dev = noise.device
        pair = self.cfg_pair
        with torch.no_grad():
            if pair is not None:                                              # same noise on both ranks of the pair
                noise = pair.broadcast(noise.contiguous())
                if initial_latent is not None:
                    initial_latent = pair.broadcast(initial_latent.to(device=dev, dtype=noise.dtype).contiguous())
            # each branch needs only its own prompt embedding (the T5 runs once per rank in a CfgPair)
            conditional_dict = (self.text_encoder(text_prompts=text_prompts) if pair is None or pair.role == 0 else None)
            unconditional_dict = (self.text_encoder(text_prompts=[self.args.negative_prompt] * len(text_prompts))
                                  if pair is None or pair.role == 1 else None)
            if getattr(self.generator_cond, "model_type", "t2v") == "i2v":
                if image_condition is None:
                    raise ValueError("the Wan-I2V model type needs image_condition = {'clip_fea', 'y'}")
                ic = {k: image_condition[k].to(device=dev, dtype=torch.bfloat16).contiguous() for k in ("clip_fea", "y")}
                if pair is not None:                                          # one image per pair: role 0's
                    ic = {k: pair.broadcast(v) for k, v in ic.items()}
                for d in (conditional_dict, unconditional_dict):
                    if d is not None:
                        d.update(ic)

            output = torch.zeros_like(noise)
            want_pos, want_neg = pair is None or pair.role == 0, pair is None or pair.role == 1
            if self.kv_cache_pos is None and self.kv_cache_neg is None:
                if want_pos:
                    self.kv_cache_pos = self.generator_cond.new_kv_cache()
                    self.crossattn_cache_pos = self.generator_cond.new_crossattn_cache()
                if want_neg:
                    self.kv_cache_neg = self.generator_cond.new_kv_cache()
                    self.crossattn_cache_neg = self.generator_cond.new_crossattn_cache()
            else:
                for c in (self.crossattn_cache_pos, self.crossattn_cache_neg):
                    for blk in (c or []):
                        blk["is_init"] = False
                for kv in (self.kv_cache_pos, self.kv_cache_neg):
                    if kv is not None:
                        kv.reset()
            live_kv = [kv for kv in (self.kv_cache_pos, self.kv_cache_neg) if kv is not None]
            for kv in live_kv:
                kv.enable_attn_history(self.attn_history)

            stages = self.plan.stages
            S = self.frame_seq_length
            first = 0
            if initial_latent is not None:
                n_init = initial_latent.shape[1]
                if self.mode == "t2v":                                        # t2v :407-439
                    groups = [(stages[0], initial_latent)]
                    first = 1
                else:                                                         # i2v :368-435
                    groups = [(stages[j], initial_latent[:, j:j + 1]) for j in range(n_init)]
                    first = n_init
                for frames, lat in groups:
                    lat = lat.to(device=dev, dtype=noise.dtype).contiguous()
                    t0 = torch.zeros([1, len(frames)], device=dev, dtype=torch.float32)
                    self._refresh(lat, conditional_dict, unconditional_dict, t0, frames)
                    output[:, frames] = lat
            elif self.mode == "i2v":
                raise ValueError("I2V needs the VAE-encoded image as initial_latent")

            for si in range(first, len(stages)):
                frames = stages[si]
                if self.handoff_poll is not None and si > self.plan.handoff_stage:
                    torch.cuda.synchronize(dev)                               # host in step with the GPU: the previous stage is done
                    self.handoff_poll()
                latents = noise[:, frames].contiguous()
                if self.plan.renoised_frames(si):                             # :279-326
                    src = (3, 10) if si == 2 else (12, 19)
                    for pos, s in ((0, src[0]), (-1, src[1])):
                        sl = slice(0, 1) if pos == 0 else slice(-1, None)
                        fresh = (torch.randn_like(latents[:, sl]) if self.renoise_override is None else
                                 self.renoise_override[frames[pos]].to(latents).unsqueeze(1))
                        if pair is not None:
                            fresh = pair.broadcast(fresh.contiguous())        # one RNG draw per pair, not per rank
                        latents[:, sl] = self.ddpm_scheduler.add_noise(
                            output[:, s:s + 1].flatten(0, 1), fresh.flatten(0, 1),
                            self.ddmp_timestep.flatten(0, 1)).unflatten(0, (1, 1))
                if self.plan.hides_anchors(si):
                    for cache in live_kv:
                        for v in (20 * S, 19 * S):
                            if v in cache.vis:
                                cache.vis.remove(v)
                elif self.plan.shows_anchors(si):
                    for cache in live_kv:
                        for v in (20 * S, 19 * S):
                            if v not in cache.vis:
                                cache.vis.append(v)

                sample_scheduler = self._initialize_sample_scheduler(noise)
                flow = torch.empty((2,) + tuple(latents.shape), device=dev, dtype=latents.dtype)   # [cond, uncond]
                branches = self._branches(conditional_dict, unconditional_dict)
                # in a CfgPair the forward writes a private buffer and the all-gather fills both halves of `flow`
                outs = [flow[i] for _, _, _, i in branches] if pair is None else [torch.empty_like(latents)]
                timestep = torch.empty([1, len(frames)], device=dev, dtype=torch.float32)
                graphs, step_graph = None, None
                if self.use_graphs:
                    starts = [f * S for f in frames]
                    timestep.fill_(float(sample_scheduler.timesteps[0]))
                    graphs = [self.generator_cond.capture(latents, d, timestep, kv, cross, starts, o)
                              for (d, kv, cross, _), o in zip(branches, outs)]
                    if pair is None and self.step_graphs:
                        # ONE hipGraph per denoise step: both forwards + the fused CFG / UniPC update, whose scalars and
                        # the next timestep live in device tables -- 50 replays with no host work in between
                        sample_scheduler.build_step_table(self.args.guidance_scale, dev)
                        sample_scheduler._ensure_state(latents)
                        torch.cuda.synchronize(dev)
                        engine = self.generator_cond.engine
                        concurrent = self.concurrent_cfg
                        if concurrent is None:
                            from ..stage_plan import concurrent_cfg_pays
                            concurrent = concurrent_cfg_pays(len(frames) * S, engine.dim)
                        ws2 = None
                        share = engine.shared_block0_buffer(len(frames)) if (self.share_block0 and not concurrent) else None
                        if concurrent:                                      # (allocations and the stream: outside the capture)
                            ws2 = engine.second_workspace(len(frames))
                            if self._side_stream is None:
                                self._side_stream = torch.cuda.Stream(device=dev)
                            torch.cuda.synchronize(dev)
                        step_graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(step_graph):
                            if concurrent:
                                main, side = torch.cuda.current_stream(dev), self._side_stream
                                (d0, kv0, cr0, _), (d1, kv1, cr1, _) = branches
                                side.wait_stream(main)                          # fork
                                self._forward(latents, d0, timestep, kv0, cr0, frames, outs[0])
                                with torch.cuda.stream(side):
                                    self._forward(latents, d1, timestep, kv1, cr1, frames, outs[1], workspace=ws2)
                                main.wait_stream(side)                          # join: the CFG / UniPC update needs both
                            else:
                                for bi, ((d, kv, cross, _), o) in enumerate(zip(branches, outs)):
                                    self._forward(latents, d, timestep, kv, cross, frames, o, share_out=share if bi == 0 else None,
                                                  share_in=share if bi == 1 else None)
                            sample_scheduler.step_cfg_table(flow[0], flow[1], latents, timestep)
                for kv in live_kv:                                            # a new stage: other shapes, other blocks
                    kv.reset_attn_history()
                if step_graph is not None:
                    timestep.fill_(float(sample_scheduler.timesteps[0]))
                    for _ in sample_scheduler.timesteps:
                        step_graph.replay()
                else:
                    for t in sample_scheduler.timesteps:
                        timestep.fill_(float(t))
                        if graphs is not None:
                            for g in graphs:
                                g.replay()
                        else:
                            for (d, kv, cross, _), o in zip(branches, outs):
                                self._forward(latents, d, timestep, kv, cross, frames, o)
                        if pair is not None:
                            pair.exchange(outs[0], flow)
                        # flow = uncond + g (cond - uncond); latents = scheduler.step(flow)  -- one fused kernel (:366-374)
                        sample_scheduler.step_cfg(flow[0], flow[1], self.args.guidance_scale, latents)

                output[:, frames] = latents
                bad = self.generator_cond.engine.share_check_failures()       # MMPL_CHECK_SHARE=1 (debug; 0 without it, no sync)
                if bad:
                    raise RuntimeError(f"MMPL_CHECK_SHARE: {bad} uncond forward(s) of stage {si} took block 0's self-attention from the cond "
                                       "forward although the layer-0 K / V of the two caches differ")
                if si == self.plan.handoff_stage:                             # t2v :380-383, i2v :340-343
                    save_latents = (torch.cat([output[:, :1], latents], dim=1) if self.mode == "t2v"
                                    else torch.cat([output[:, :1], output[:, -2:]], dim=1))
                    if pair is not None and pair.role != 0:
                        pass                                                  # the cond rank of the pair delivers it
                    elif self.handoff_sink is not None:
                        self.handoff_sink(save_latents)
                    elif self.save:
                        torch.save(save_latents, self.save)
                # the stage that does not persist its K/V ([13..18]) gains nothing from the refresh pass: the reference
                # runs it anyway (casual_fps_inference.py:385-403) but it writes no cache and its output is dropped
                if self.plan.write_slots(frames)[0] < 0:
                    pass
                elif graphs is not None:                                        # refresh pass = same graphs at t = 0
                    timestep.zero_()
                    for g in graphs:
                        g.replay()
                else:
                    self._refresh(latents, conditional_dict, unconditional_dict, timestep, frames)

            video = None
            if decode and (pair is None or pair.role == 0):
                video = self.vae.decode_to_pixel(output)
                video = (video * 0.5 + 0.5).clamp(0, 1)
        if return_latents:
            return video, output
        return video

    def _initialize_sample_scheduler(self, noise):
        s = FlowUniPCMultistepScheduler(num_train_timesteps=self.num_train_timesteps, shift=1, use_dynamic_shifting=False)
        s.set_timesteps(self.sampling_steps, device=noise.device, shift=self.shift)
        self.timesteps = s.timesteps
        return s
