#!/usr/bin/env python3
"""Headline benchmark: chunk-AR denoising throughput of Wan2.1-T2V-14B at 720p latent shapes on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one denoise step of the reference's hot loop (MMPL_t2v/pipeline/casual_fps_inference.py:338-374):
cond DiT forward + uncond DiT forward (separate KV caches) + CFG combine + UniPC update, all HIP.  Steps rotate
through the four T2V stage shapes s0..s3 (2/7/6/6 query frames attending 2/9/13/21 frames), so K steps (any K; each step is also timed by a HIP event pair and the chunk time is assembled per stage)
sample the stages in turn.  A first chunk is 4 stages x (50 denoise steps + 1 cache-refresh pair) = 204
step-equivalents (408 forwards), hence

    latent-frames/s = 21 / (51 * sum over the 4 stages of that stage's mean step time)
                                                           (whole-job: summed over ranks, each rank = one chunk)

Inputs (weights, KV caches, context, latents) are synthetic and resident in HBM before the timed region.

What is timed: by default every step is ONE hipGraph replay (both forwards + the fused CFG / UniPC update with its scalars
in device tables: the north star's "hipGraph capture of one denoise step", exactly what
mmpl_amd.pipeline.CausalFPSInferencePipeline replays 50 times per stage).  `--eager` times plain launches instead.  After
the timed region one more rotation of the four stages runs eagerly with a hipEvent pair around every self-attention
launch on the launch stream: that pass gives the `eager` figures printed beside the headline and the `roofline` object
(per-kernel events cannot be recorded inside a graph replay; the kernels are the same binaries with the same arguments,
and the rocprofv3 kernel trace of this command, profiles/, covers both passes).  `cpu_baseline` = the oracle (a CPU port)
timed on a bounded sample: one transformer block per stage shape.

N > 1: one process per GPU, and the default line is a MEASUREMENT of the path the reference's multi-GPU scripts run: ONE video of
C = 2 x lanes chunks through the real pipeline and the real dependency chain (chunk c on lane c % lanes, anchors handed lane ->
lane + 1 over RCCL after the anchor stage; from 4 ranks on the ranks are paired cond | uncond, the reference's device_cond /
device_uncond seam, `--no-cfg-split` / `--cfg-split` override).  `value` = 21 C / wall (first noise -> last latent).  To fit a
driver run, the UniPC steps per stage are chosen from a wall-clock budget (`--wavefront-budget-s`, `--sampling-steps 50` = the
reference's); every stage's time is proportional to its forwards (2 K + 2), so the shortened run's wall is scaled by 102 / (2 K + 2)
and both numbers are printed (`value` scaled, `value_shortened_run` raw, `value_modelled` = the occupancy model).  `--rotation`
prints the K-rotating-steps line of N = 1 on every rank instead (`value` = lanes the wavefront can keep busy x 21 / chunk time:
a model, labelled so).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required for RCCL between processes on this pool

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
KIND_NAMES = ["gemm", "attn_self", "attn_cross", "layernorm", "qknorm_rope", "elementwise", "cfg_unipc", "vae"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one process per GPU).  Inside a torch.distributed.run launch it must equal WORLD_SIZE; outside one, "
                         "N > 1 makes this process start that launch itself (before it touches the GPU) and relay rank 0's JSON line")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one rank per GPU (the measured configuration).  gloo: functional runs of the N > 1 code "
                         "paths on a box with fewer GPUs than ranks (ranks share devices, exchanges are staged through the host) -- "
                         "the JSON line says so and is not a scaling measurement")
    ap.add_argument("--dry-run-launch", action="store_true",
                    help="launcher self-test without GPUs: every rank joins a gloo group, rank 0 prints how many ranks it saw")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--model", default="14B", choices=["14B", "1.3B", "small", "tiny"])
    ap.add_argument("--res", default="720p", choices=["480p", "720p", "tiny"])
    ap.add_argument("--mode", default="t2v", choices=["t2v", "i2v"], help="stage plan of the rotation (i2v: 1/7/6/6 query frames, BASELINE configs[4])")
    ap.add_argument("--i2v-model", action="store_true",
                    help="Wan-I2V MODEL TYPE (in_dim 36 + the CLIP image stream in every block's cross-attention, model.py:563-616) instead of the "
                         "T2V backbone the reference's I2V scripts run; synthetic CLIP features; not the headline workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="disable the per-kernel hipEvent pairs")
    ap.add_argument("--profile-all", action="store_true",
                    help="time every kernel class (adds gemm_tflops / kernel_time_share; ~0.5-1 %% slower than the default, "
                         "which times the roofline kernel -- self-attention -- only)")
    ap.add_argument("--no-vae", action="store_true", help="skip the (untimed-region) VAE decode measurement")
    ap.add_argument("--cfg-split", action="store_true",
                    help="N/2 chunk lanes x (cond, uncond) rank pairs exchanging flow predictions every step "
                         "(the reference's device_cond/device_uncond seam) instead of N chunk lanes; default from 4 ranks on")
    ap.add_argument("--no-cfg-split", action="store_true", help="N chunk lanes whatever N is")
    ap.add_argument("--eager", action="store_true", help="time plain launches instead of one hipGraph replay per step")
    ap.add_argument("--concurrent-cfg", action="store_true",
                    help="ALWAYS the cond and the uncond forward of a step as two PARALLEL branches of the step graph (a second stream, a private "
                         "workspace): the tail of one branch's kernels (partial last rounds, split tails, launch gaps) is filled by the other's.  "
                         "Default: per stage where it pays (mmpl_amd.stage_plan.concurrent_cfg_pays -- what the pipeline does)")
    ap.add_argument("--no-concurrent-cfg", action="store_true", help="never: the two forwards one after the other on one stream")
    ap.add_argument("--no-share-block0", action="store_true",
                    help="the uncond forward recomputes block 0's self-attention instead of taking x after it from the cond forward (the two "
                         "branches run it on identical inputs; what the pipeline's sequential step graphs do; bit-identical either way)")
    ap.add_argument("--wavefront-chunks", type=int, default=0,
                    help="C > 0: instead of K rotating steps, run ONE video of C chunks through the real pipeline and the real dependency chain "
                         "(chunk c on lane c %% lanes, RCCL anchor hand-off after the anchor stage, VAE consumer transform) and report the MEASURED "
                         "wall clock first noise -> last latent: value = 21 C / wall (SURVEY 8d), per-rank busy fraction, stagger, hand-off latency")
    ap.add_argument("--sampling-steps", type=int, default=None, help="UniPC steps per stage of the measured wavefront (50 = the reference's and the "
                         "default on one rank; N > 1 default: chosen from --wavefront-budget-s; fewer than 50 = a shortened run whose wall clock is "
                         "scaled by 102 / (2 K + 2) and labelled so)")
    ap.add_argument("--wavefront-budget-s", type=float, default=420.0, help="N > 1 default line: wall-clock budget of the measured wavefront's timed region")
    ap.add_argument("--rotation", action="store_true", help="N > 1: the K-rotating-steps line (occupancy MODEL for `value`) instead of the measured wavefront")
    ap.add_argument("--attn-stats", action="store_true", help="count the self-attention blocks the FAST softmax pass could not hold (one atomic per "
                         "256-row block inside the timed region; always on with --heavy-tail)")
    ap.add_argument("--probe-seconds", type=float, default=1.5, help="same-run calibration: seconds of back-to-back MFMAs per instruction shape right "
                         "before the warm-up (0 = skip)")
    ap.add_argument("--heavy-tail-gain", type=float, default=8.0, help="QK-norm gain multiplier of --heavy-tail (8 = the test's; logit std ~ gain^2 nats)")
    ap.add_argument("--heavy-tail", action="store_true",
                    help="NOT the headline: the statistics real checkpoints have and unit-variance synthetic weights do not (QK-norm gains x8, "
                         "six massive-activation channels -- the weights of tests/test_fullsize_gpu.py's heavy-tail case): how much of the "
                         "self-attention kernel's speed is its max-free FAST pass?  Reports attn_blocks_redone_fraction next to the attention ms")
    ap.add_argument("--heavy-tail-heads", type=float, default=None, metavar="P",
                    help="with --heavy-tail: the QK-norm gain only on a fraction P of every layer's heads (their 128 norm channels; a seeded draw "
                         "per layer) -- closer to a real checkpoint than every head x gain; the massive-activation channels stay")
    ap.add_argument("--no-attn-history", action="store_true",
                    help="A/B: stateless self-attention (no per-block pass history; every block whose FAST pass fails pays for both passes every step)")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0, help="stop adding stage shapes to the CPU baseline after this many seconds")
    return ap.parse_args()


def cpu_baseline(cfg, lat_h, lat_w, stage_shapes, budget_s):
    """Oracle (CPU port of the reference path) on a bounded sample: ONE transformer block of the benchmarked model per
    stage shape, all host threads; s0 is run 3 times after a warm-up, the larger shapes once each while the
    time budget lasts (a shape not reached is priced at the FLOP rate of the measured ones).  Reported in the metric's unit:
    chunk = 102 forwards per stage x num_layers blocks (a chunk is ~481 PFLOP at 14B/720p -- hours on a CPU)."""
    from mmpl_amd.synthetic import dit_state_dict
    from oracle import wan_dit_ref as W
    one = dict(cfg, num_layers=1)
    ocfg = W.DitCfg(**one)
    sd = dit_state_dict(one, seed=0)
    gh, gw = lat_h // 2, lat_w // 2
    S = gh * gw
    freqs = W.rope_table(128)
    ctx = torch.randn(1, 512, ocfg.dim).to(torch.bfloat16)
    ck, cv = W.cross_kv(sd, ocfg, 0, ctx)
    t_start, secs, flops, notes = time.time(), [], [], []
    for si, (nq, nkv) in enumerate(stage_shapes):
        Lq = nq * S
        fl = 2.0 * Lq * (6.0 * ocfg.dim ** 2 + 2.0 * ocfg.dim * ocfg.ffn_dim) + 4.0 * Lq * (nkv * S) * ocfg.dim + 4.0 * Lq * 512 * ocfg.dim
        flops.append(fl)
        if si > 0 and time.time() - t_start + (fl / (sum(flops[:len(secs)]) / sum(secs))) > budget_s:
            continue
        persist = nkv <= nq or si < 3                  # the last T2V / I2V stage attends its own K/V without persisting them
        n_slots = nkv if persist else nkv - nq
        kv = W.new_kv_cache(ocfg, max(n_slots, nq), S)[0]
        kv["k"].normal_()
        kv["v"].normal_()
        x = torch.randn(1, Lq, ocfg.dim).to(torch.bfloat16)
        e0 = (torch.randn(1, nq, 6, ocfg.dim) * 0.1).to(torch.bfloat16)
        frames = list(range(nq))
        ws = frames if persist else [-1] * nq
        vis = list(range(nkv)) if persist else list(range(nkv - nq))
        reps = 3 if si == 0 else 1
        with torch.no_grad():
            if si == 0:
                W.block_forward(sd, ocfg, 0, x, e0, kv, ck, cv, frames, ws, vis, S, gh, gw, freqs)   # untimed: thread pool / allocator warm-up
            t0 = time.time()
            for _ in range(reps):
                W.block_forward(sd, ocfg, 0, x, e0, kv, ck, cv, frames, ws, vis, S, gh, gw, freqs)
        dt = (time.time() - t0) / reps
        secs.append(dt)
        notes.append(f"s{si} (Lq={Lq}, Lkv={nkv * S}) {dt:.1f} s x{reps}")
        del kv, x
    rate = sum(flops[:len(secs)]) / sum(secs)
    stage_block_s = secs + [f / rate for f in flops[len(secs):]]
    chunk_s = 102.0 * cfg["num_layers"] * sum(stage_block_s)
    return {"value": 21.0 / chunk_s, "unit": "latent-frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle block_forward, 1 of {cfg['num_layers']} blocks per stage shape: " + ", ".join(notes) +
                      (f"; {len(flops) - len(secs)} larger shape(s) priced at the measured {rate / 1e12:.2f} TFLOP/s (time budget)" if len(secs) < len(flops) else
                       f"; {rate / 1e12:.2f} TFLOP/s") + "; a chunk = 102 forwards per stage x all blocks"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a torch.distributed.run launch: become that launch.  This process has not touched
    the GPU (import torch does not), and it never execs: the ranks are CHILD processes (one per GPU, RCCL between them),
    their stdout is ours (rank 0 prints the one JSON line), and a failing rank makes the whole command fail."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def dry_run_launch(world: int, rank: int):
    """Every rank joins a gloo group and contributes 1; rank 0 reports the sum (what `n_gpus` would be)."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ.get("MMPL_DRY_RUN_FAIL_RANK") == str(rank):       # launcher test: a rank that dies must fail the command
        sys.exit(7)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": dist.get_world_size(), "ranks_seen": int(t.item())}), flush=True)
    dist.destroy_process_group()


def run_wavefront(args, dist, rank, world, dev, gloo):
    """`--wavefront-chunks C`: one video of C chunks, measured.  The real pipeline (mmpl_amd.pipeline: stage loop, hipGraph per
    denoise step, hand-off sink after the anchor stage), the real dependency chain (run_chunk_wavefront: chunk c on lane c % lanes
    waits for chunk c-1's anchors, turns them into its two initial latents with the VAE prefix decode / encode) and the real
    exchange (ChunkHandoff: RCCL p2p on a side stream behind the host-side ready handshake; 2-rank CFG pairs from 4 ranks on).
    Reference: Wan_fps_inference_parallel_4gpu_20s.py:180-261 (threads + .pt files + 1 Hz poll), ..._5-60s.py:188-381."""
    import types
    from mmpl_amd.geometry import RESOLUTIONS, Geometry
    from mmpl_amd.handoff import CfgPair, ChunkHandoff, handoff_to_initial_latent, run_chunk_wavefront
    from mmpl_amd.pipeline import CausalFPSInferencePipeline
    from mmpl_amd.stage_plan import T2V_STAGE_SHAPES, dit_forward_flops
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict, vae_state_dict
    from mmpl_amd.wan_wrapper import SyntheticTextEncoder, WanFPSWrapper, WanVAEWrapper
    torch.set_grad_enabled(False)
    cfg = WAN_CONFIGS[args.model]
    lat_h, lat_w = (16, 24) if args.res == "tiny" else RESOLUTIONS[args.res]
    geo = Geometry(lat_h, lat_w)
    cfg_split = args.cfg_split or (not args.no_cfg_split and world >= 4 and world % 2 == 0)
    pair, heads, lay = (None, None, None)
    if world > 1 and cfg_split:
        pair, heads, lay = CfgPair.build(world, dev, True)
    n_lanes = world // 2 if pair is not None else world
    lane = lay["lane_of"][rank] if lay else rank
    C_ = args.wavefront_chunks if args.wavefront_chunks > 0 else 2 * n_lanes
    gen = WanFPSWrapper("Wan2.1-T2V-14B", timestep_shift=5.0, is_causal=True, model_config=cfg, geometry=geo, device=str(dev))
    gen.load_state_dict(dit_state_dict(cfg, seed=1234, device=dev))
    enc = SyntheticTextEncoder(cfg.get("text_dim", 4096), str(dev))
    vae = WanVAEWrapper(geometry=geo, device=str(dev), state_dict=vae_state_dict(seed=7))
    pargs = types.SimpleNamespace(model_kwargs={}, num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0, negative_prompt="",
                                  independent_first_frame=False, sampling_steps=args.sampling_steps or 50)
    pipe = CausalFPSInferencePipeline(pargs, str(dev), generator=gen, text_encoder=enc, vae=vae, save=None, mode="t2v", geometry=geo)
    pipe.cfg_pair = pair
    ho = None
    if world == 1:
        ho = None
    elif pair is None or pair.role == 0:
        ho = ChunkHandoff((1, 8, 16, lat_h, lat_w), dev, group=heads)
    g = torch.Generator(device="cpu").manual_seed(0)
    noises = [torch.randn([1, 21, 16, lat_h, lat_w], generator=g).to(torch.bfloat16) for _ in range(C_)]
    # untimed warm-up: one 1-step chunk and the consumer transform compile / allocate everything (graphs are re-captured per stage)
    pipe.sampling_steps = 1
    _, lat = pipe.inference(noises[0].to(dev), ["warm-up"], return_latents=True, decode=False)
    handoff_to_initial_latent(vae, torch.cat([lat[:, :1], lat[:, [2, 3, 10, 11, 12, 19, 20]]], dim=1))
    torch.cuda.synchronize()
    if args.sampling_steps is None:
        # UniPC steps per stage from the wall-clock budget: a second 1-step chunk (4 forwards per stage, everything warm) is timed,
        # a K-step chunk costs (2 K + 2) / 4 of it, and the wavefront takes C / lanes chunk times + (lanes - 1) staggers of ~0.3 chunk
        tw = time.time()
        pipe.inference(noises[0].to(dev), ["warm-up"], return_latents=True, decode=False)
        torch.cuda.synchronize()
        one = torch.tensor([time.time() - tw], dtype=torch.float64)
        if dist is not None and world > 1:
            one = one.to("cpu" if gloo else dev)
            dist.all_reduce(one, op=dist.ReduceOp.MAX)
        chunks_of_wall = (C_ + n_lanes - 1) // n_lanes + 0.3 * (n_lanes - 1)
        k = int((args.wavefront_budget_s / (chunks_of_wall * one.item() / 4.0) - 2.0) / 2.0)
        args.sampling_steps = max(2, min(50, k)) if world > 1 else 50
    pipe.sampling_steps = args.sampling_steps
    torch.cuda.synchronize()
    marks = {}

    def make_chunk(c, initial, sink):
        m = marks.setdefault(c, {})
        m["t_start"] = time.time()

        def tee(t):
            torch.cuda.synchronize()                   # the anchor stage is really done (the host would otherwise be ahead of the GPU)
            m["t_anchor"] = time.time()
            sink(t)
        pipe.handoff_sink = tee
        pipe.handoff_poll = ho.poll if ho is not None else None
        _, lat_c = pipe.inference(noises[c].to(dev), ["a cat running on the grass"], initial_latent=initial, return_latents=True, decode=False)
        torch.cuda.synchronize()
        m["t_end"] = time.time()
        return lat_c

    def to_initial(recv):
        return handoff_to_initial_latent(vae, recv.to(dev))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.time()
    if world == 1:
        initial = None
        for c in range(C_):                              # one rank: the chain is sequential (chunk c+1 needs chunk c's anchors)
            got = {}
            lat_c = make_chunk(c, initial, lambda t: got.setdefault("h", t))
            if c + 1 < C_:
                initial = to_initial(got["h"])
    else:
        run_chunk_wavefront(make_chunk, C_, ho, to_initial, gather=False, pair=pair, lane=lane, n_lanes=n_lanes,
                            initial_like=torch.empty([1, 2, 16, lat_h, lat_w], device=dev, dtype=torch.bfloat16))
    barrier()
    wall = time.time() - t0
    mine = {"rank": rank, "marks": marks, "handoff": (ho.stats if ho is not None else {}), "t0": t0}
    everyone = [mine]
    if dist is not None and world > 1:
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
    if rank != 0:
        return
    S = (lat_h // 2) * (lat_w // 2)
    stage_flops = [dit_forward_flops(cfg, S, q, kv) for q, kv in T2V_STAGE_SHAPES]
    chunks = {}
    for e in everyone:
        if lay is not None and lay["role_of"][e["rank"]] != 0:
            continue
        for c, m in e["marks"].items():
            chunks[c] = dict(m, rank=e["rank"])
    ho_stats = {}
    for e in everyone:
        for c, st in e["handoff"].items():
            ho_stats.setdefault(c, {}).update(st)
    order = sorted(chunks)
    dur = [chunks[c]["t_end"] - chunks[c]["t_start"] for c in order]
    anchor = [chunks[c].get("t_anchor", chunks[c]["t_end"]) - chunks[c]["t_start"] for c in order]
    stagger = [chunks[order[i + 1]]["t_start"] - chunks[order[i]]["t_start"] for i in range(len(order) - 1)]
    busy = {}
    for c in order:
        busy[chunks[c]["rank"]] = busy.get(chunks[c]["rank"], 0.0) + chunks[c]["t_end"] - chunks[c]["t_start"]
    lat_ho = {c: st["t_recv_done"] - max(st.get("t_sink", 0.0), st.get("t_ready", 0.0)) for c, st in ho_stats.items() if "t_recv_done" in st}
    wait_ho = {c: max(0.0, st["t_sink"] - st["t_ready"]) for c, st in ho_stats.items() if "t_sink" in st and "t_ready" in st}
    full = args.sampling_steps == 50
    # Forwards per stage at K UniPC steps (casual_fps_inference.py:266-439): 2 K + 2 (K denoise steps x (cond, uncond) + the refresh pair),
    # except the stage that does not persist its K / V ([13..18]: 2 K, no refresh pair -- the pipeline skips it, :283) and the first stage
    # of chunks >= 2 (2: the refresh pair on the handed-over frames, whatever K is).  The wall clock of the shortened run is scaled by
    # the FLOP-weighted ratio of those counts at 50 and at K over all chunks of the video -- per stage, not one 102 / (2 K + 2) for
    # everything (which over-states `value` for K < 50: ~0.6 % at K = 27, ~10 % at K = 2).  What does not shrink with K (hand-off, the
    # consumer's VAE transform) is over-weighted by the shortened run, which pulls the scaled value DOWN; the FLOP weights stand in for
    # measured stage times, whose error has no known sign: `value_shortened_run` is the measurement, `value` its projection.
    def chunk_fwd_flops(K, first):
        n = [2 * K + 2 if first else 2, 2 * K + 2, 2 * K + 2, 2 * K]
        return sum(f * k for f, k in zip(stage_flops, n))
    to_50 = (sum(chunk_fwd_flops(50, c == 0) for c in range(C_)) / sum(chunk_fwd_flops(args.sampling_steps, c == 0) for c in range(C_)))
    value_raw = 21.0 * C_ / wall
    value = value_raw / to_50
    # the occupancy model of the default N > 1 line, fed by THIS run's chunk / anchor times (kept for comparison)
    later = dur[1:] if len(dur) > 1 else dur
    later_anchor = anchor[1:] if len(anchor) > 1 else anchor
    d_mean, a_mean = sum(later) / len(later), sum(later_anchor) / len(later_anchor)
    modelled = min(float(n_lanes), d_mean / a_mean) * 21.0 / (d_mean * to_50)
    flops = sum(chunk_fwd_flops(args.sampling_steps, c == 0) for c in range(C_))
    n_fwd_steps = sum(1 for c in order for _ in range(4 if c == 0 else 3)) * (args.sampling_steps + 1)
    res = {"metric": "video_latent_frames_per_sec", "value": value, "unit": "latent-frames/s",
           "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": sum(dur) / n_fwd_steps * 1e3, "ms_per_step_note": "busy time of all chunks / their denoise-step equivalents (K + 1 per stage run); "
                                                                          "--steps / --warmup do not apply to the measured wavefront",
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           **({"functional_only": "gloo backend: ranks share GPUs, exchanges staged through the host -- exercises the N > 1 code paths, "
                                  "not a scaling measurement"} if gloo else {}),
           **({} if full else {"value_shortened_run": value_raw,
                               "value_scaling": f"the run used {args.sampling_steps} UniPC steps per stage instead of the reference's 50 (wall-clock budget "
                                                f"{args.wavefront_budget_s:.0f} s); `value` = 21 C / (wall x {to_50:.3f}): the FLOP-weighted ratio of the forwards "
                                                "every stage of every chunk runs at 50 steps and at K (2 K + 2 per stage; 2 K for the stage that does not persist "
                                                "its K / V; 2 for the first stage of chunks >= 2).  The hand-off / VAE consumer transform do not shrink with K "
                                                "(pulls `value` down); FLOP weights stand in for stage times (sign unknown): `value_shortened_run` = 21 C / wall "
                                                "is the measurement, `value` its projection to the reference's length; --sampling-steps 50 measures it"}),
           "config": {"workload": f"Wan2.1-T2V-{args.model} {args.res}: ONE video of {C_} chunks (21 latent frames each) through the real pipeline, "
                                  f"{args.sampling_steps} UniPC steps x CFG per stage, anchors handed lane -> lane + 1",
                      "frame_seqlen": S, "latent_hw": [lat_h, lat_w], "sampling_steps": args.sampling_steps, "guidance_scale": 5.0,
                      "timed_path": "wall clock first noise -> last latent over all ranks (barrier + synchronize on both sides)",
                      "parallelism": f"measured wavefront: {C_} chunks on {n_lanes} lane(s)" +
                                     (" x 2 (cond|uncond) CFG split" if pair is not None else "") +
                                     (", RCCL p2p anchor hand-off behind a host-side ready handshake" if world > 1 and not gloo else
                                      (", hand-off staged through the host (gloo)" if world > 1 else ", sequential chain on one rank"))},
           "wall_s": wall, "chunks": C_, "lanes": n_lanes,
           "wall_scale_to_50_steps": to_50,                 # `value` = 21 C / (wall x this); 1.0 at --sampling-steps 50
           "value_modelled": modelled,
           "value_modelled_note": "min(lanes, chunk_s / anchor_s) * 21 / chunk_s with this run's mean chunk and anchor-stage times of chunks >= 2 "
                                  "(the occupancy model the K-steps N > 1 line prints)",
           "chunk_s": dur, "anchor_done_after_s": anchor, "stagger_s": stagger,
           "rank_busy_fraction": {str(r): b / wall for r, b in sorted(busy.items())},
           "handoff_latency_s": {str(c): v for c, v in sorted(lat_ho.items())},
           "handoff_latency_note": "recv complete - max(anchors available on the producer, consumer ready): RCCL p2p + header (+ host staging under gloo)",
           "producer_waited_for_consumer_s": {str(c): v for c, v in sorted(wait_ho.items())},
           "achieved_pflops_all_gpus": flops / wall / 1e15}
    if not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(cfg, lat_h, lat_w, T2V_STAGE_SHAPES, args.cpu_budget_s)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU measurement over it
            res["cpu_baseline"] = {"value": None, "unit": "latent-frames/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e!r}"}
    print(json.dumps(res), flush=True)


def main():
    args = parse()
    in_launch = "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is None:
        args.gpus = world
    if not in_launch and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but this launch has WORLD_SIZE={world} ranks; n_gpus must be the ranks that run")
    if args.dry_run_launch:
        return dry_run_launch(world, rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # (a 1-rank torchrun launch exercises the RCCL init / barrier / all-reduce path too)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # the wavefront's receives wait for whole anchor stages (minutes at 14B/720p): never the 10-minute default
        if args.dist_backend == "gloo":
            local_rank %= max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo", timeout=datetime.timedelta(hours=4))
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"), timeout=datetime.timedelta(hours=4))
    assert args.steps > 0 and args.warmup >= 0
    gloo = dist is not None and args.dist_backend == "gloo"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if args.wavefront_chunks > 0 or (world > 1 and not args.rotation):
        run_wavefront(args, dist, rank, world, dev, gloo)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    pair, lanes, lane, n_lanes = None, None, rank, world
    if not args.cfg_split and not args.no_cfg_split and world >= 4 and world % 2 == 0:
        args.cfg_split = True            # more chunk lanes than the wavefront can keep busy: spend the ranks on the CFG pair instead
    if args.cfg_split:
        assert world >= 2 and world % 2 == 0, "--cfg-split needs an even number of ranks"
        from mmpl_amd.handoff import CfgPair
        pair, _, lay = CfgPair.build(world, dev, cfg_split=True)
        lanes, lane, n_lanes = lay["heads"], lay["lane_of"][rank], world // 2

    from mmpl_amd import _lib
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.geometry import RESOLUTIONS
    from mmpl_amd.scheduler import FlowUniPCMultistepScheduler
    from mmpl_amd.stage_plan import I2V_STAGE_SHAPES, N_SLOTS, T2V_STAGE_SHAPES, StagePlan, dit_forward_flops, slot_of
    from mmpl_amd.synthetic import WAN_CONFIGS, dit_state_dict

    lib = _lib.load()
    cfg = WAN_CONFIGS[args.model]
    lat_h, lat_w = (16, 24) if args.res == "tiny" else RESOLUTIONS[args.res]
    if args.i2v_model:
        from mmpl_amd.synthetic import dit_i2v_state_dict
        cfg = dict(cfg, model_type="i2v")
        eng = DitEngine(cfg, lat_h, lat_w, dev)
        eng.load_state_dict(dit_i2v_state_dict(cfg, seed=1234, device=dev))
        eng.set_image_kv(*eng.precompute_image_context(torch.randn(257, 1280, device=dev).to(torch.bfloat16)))
    else:
        eng = DitEngine(cfg, lat_h, lat_w, dev)
        sd = dit_state_dict(cfg, seed=1234, device=dev)
        if args.heavy_tail:
            H = cfg["num_heads"]
            for l in range(cfg["num_layers"]):
                mult = torch.full([H], args.heavy_tail_gain)
                if args.heavy_tail_heads is not None:                 # only round(P * H) heads of this layer (seeded per layer)
                    gsel = torch.Generator().manual_seed(4321 + l)
                    heavy = torch.randperm(H, generator=gsel)[:max(0, min(H, round(args.heavy_tail_heads * H)))]
                    mult = torch.ones(H)
                    mult[heavy] = args.heavy_tail_gain
                mult = mult.repeat_interleave(128).to(dev)
                for k in ("self_attn.norm_q.weight", "self_attn.norm_k.weight"):
                    sd[f"blocks.{l}.{k}"] = (sd[f"blocks.{l}.{k}"].float() * mult).to(torch.bfloat16)
            b = sd["patch_embedding.bias"].float()
            cols = [c % cfg["dim"] for c in (7, 300, 1111, 2049, 3333, 5000)]
            b[cols] = torch.tensor([60.0, -60.0, 45.0, -45.0, 60.0, -50.0], device=b.device)
            sd["patch_embedding.bias"] = b.to(torch.bfloat16)
        eng.load_state_dict(sd)
        del sd
    # {query blocks, blocks the FAST softmax pass could not hold}: one atomic per block, captured into the step graphs -- diagnostic runs only
    attn_stats = eng.enable_attn_stats() if (args.attn_stats or args.heavy_tail) else None
    S = eng.S
    plan = StagePlan(args.mode)
    stage_shapes = T2V_STAGE_SHAPES if args.mode == "t2v" else I2V_STAGE_SHAPES

    # two KV caches (cond / uncond) filled with unit-variance data so softmax sees realistic score spreads
    caches = []
    for i in range(2 if pair is None else 1):                 # a CFG pair rank owns one branch only
        kc, vc = eng.new_kv_cache(N_SLOTS)
        kc.normal_()
        vc.normal_()
        ctx = torch.randn(512, cfg["text_dim"], device=dev).to(torch.bfloat16)
        ctx[64:] = 0
        ckv = eng.precompute_context(ctx)
        # (.., cross_rows): the padded tail of the text K / V is one repeated row from row `rows` on -- explicit data, the same
        # path CrossAttnCache takes in the pipeline
        if i == 1:
            # what the pipeline's caches satisfy by construction (every forward runs on both branches with the same latents and timestep):
            # layer 0 holds the same K / V in both -- the precondition of sharing block 0's self-attention between the branches
            kc[0].copy_(caches[0][0][0])
            vc[0].copy_(caches[0][1][0])
        caches.append((kc, vc, ckv[0], ckv[1], ckv.rows))
    # per-stage state: latents, visible slots, scheduler
    vis_frames = [[0, 1], [0, 1, 2, 3, 10, 11, 12, 19, 20], list(range(13)) + ([] if args.mode == "t2v" else [19, 20]),
                  list(range(13)) + [19, 20]]
    stage_state = []
    for si, frames in enumerate(plan.stages if args.mode == "t2v" else plan.stages[1:]):
        lat = torch.randn(len(frames), 16, lat_h, lat_w, device=dev).to(torch.bfloat16)
        sched = FlowUniPCMultistepScheduler(1000, 2, 1.0)
        sched.set_timesteps(50, shift=5.0)
        vis = [slot_of(f) for f in vis_frames[si]]
        if pair is not None:
            pair.broadcast(lat)
        flow = torch.empty((2,) + tuple(lat.shape), device=dev, dtype=lat.dtype)
        x36 = None
        if args.i2v_model:                                  # x = [latents | conditioning video y (mask + image latents, 20 channels)]
            x36 = torch.randn(len(frames), 36, lat_h, lat_w, device=dev).to(torch.bfloat16)
        # the self-attention's pass history: one per (stage, CFG branch), like the pipeline's (it keeps one per branch and zeroes it per
        # stage; the rotation here interleaves the stages, so each keeps its own)
        hist = [None, None] if args.no_attn_history else [eng.new_attn_history(len(frames)) for _ in range(2 if pair is None else 1)]
        stage_state.append(dict(frames=frames, lat=lat, x36=x36, sched=sched, vis=vis, ws=plan.write_slots(frames), hist=hist,
                                fc=flow[0], fu=flow[1], flow=flow, mine=torch.empty_like(lat),
                                t=torch.empty(len(frames), dtype=torch.float32, device=dev)))
    handoff_send = torch.zeros(8, 16, lat_h, lat_w, device=dev, dtype=torch.bfloat16)
    handoff_recv = torch.zeros_like(handoff_send)
    side = torch.cuda.Stream(device=dev)

    def xin(st):
        """the forward's input: the latents, or (I2V model type) the 36-channel buffer whose first 16 channels are refreshed from them"""
        if st["x36"] is None:
            return st["lat"]
        st["x36"][:, :16].copy_(st["lat"])
        return st["x36"]

    def one_step(i, eager=False):
        st = stage_state[i % 4]
        sched = st["sched"]
        if st.get("graph") is not None and not eager:
            # ONE hipGraph per denoise step: both forwards + CFG / UniPC with device-resident scalars; the device step counter
            # and the timestep tensor advance inside the graph, the host only rewinds them after the 50th replay
            if st["replays"] >= 50:
                sched.reset_step_table(st["t"])
                st["replays"] = 0
            st["graph"].replay()
            st["replays"] += 1
            if i % 4 == 1:
                handoff_exchange(st)
            return
        if sched.step_index >= 50:
            sched.set_timesteps(50, shift=5.0)
        st["t"].fill_(float(sched.timesteps[sched.step_index]))
        if pair is None:
            for which, out in ((0, st["fc"]), (1, st["fu"])):
                kc, vc, ck, cv, crows = caches[which]
                eng.forward(xin(st), st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=out, cross_rows=crows, attn_history=st["hist"][which])
        else:
            kc, vc, ck, cv, crows = caches[0]
            if st.get("fwd_graph") is not None and not eager:
                xin(st)                                # (refreshes the 36-channel buffer of the i2v model type)
                st["fwd_graph"].replay()               # this rank's branch: one hipGraph per forward
            else:
                eng.forward(xin(st), st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=st["mine"], cross_rows=crows, attn_history=st["hist"][0])
            pair.exchange(st["mine"], st["flow"])      # host-issued 2-rank all-gather: the step cannot be ONE graph here
        sched.step_cfg(st["fc"], st["fu"], 5.0, st["lat"])
        if i % 4 == 1:
            handoff_exchange(st)

    def handoff_exchange(st):
        """chunk hand-off of the anchor stage (casual_fps_inference.py:380-383 -> RCCL p2p on a side stream, overlapped with
        the next stage's compute): lane l -> lane l+1"""
        if dist is None or world < 2 or (pair is not None and pair.role != 0):
            return
        handoff_send[0].copy_(st["lat"][0])
        handoff_send[1:].copy_(st["lat"])
        nxt = rank + 1 if lanes is None else (lanes[lane + 1] if lane + 1 < n_lanes else world)
        prv = rank - 1 if lanes is None else (lanes[lane - 1] if lane > 0 else -1)
        if gloo:                                   # functional path: through the host, on the compute stream
            reqs, rbuf = [], None
            if nxt < world:
                reqs.append(dist.isend(handoff_send.cpu(), nxt))
            if prv >= 0:
                rbuf = torch.empty(handoff_recv.shape, dtype=handoff_recv.dtype)
                reqs.append(dist.irecv(rbuf, prv))
            for w in reqs:
                w.wait()
            if rbuf is not None:
                handoff_recv.copy_(rbuf)
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops = []
            if nxt < world:
                ops.append(dist.P2POp(dist.isend, handoff_send, nxt))
            if prv >= 0:
                ops.append(dist.P2POp(dist.irecv, handoff_recv, prv))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    use_graph = pair is None and not args.eager           # (the CFG pair's per-step all-gather is a host-issued RCCL call)
    graphed = not args.eager                              # some hipGraph is replayed in the timed region -> per-kernel events come from the eager pass
    if use_graph:
        for st in stage_state:                              # eager warm-up of every launch shape, then capture
            for which, out in ((0, st["fc"]), (1, st["fu"])):
                kc, vc, ck, cv, crows = caches[which]
                st["t"].fill_(999.0)
                eng.forward(xin(st), st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=out, cross_rows=crows, attn_history=st["hist"][which])
            sched = st["sched"]
            sched.build_step_table(5.0, dev)
            sched._ensure_state(st["lat"])
            st["t"].fill_(float(sched.timesteps[0]))
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            from mmpl_amd.stage_plan import concurrent_cfg_pays
            st["concurrent"] = (args.concurrent_cfg or concurrent_cfg_pays(len(st["frames"]) * S, eng.dim)) and not args.no_concurrent_cfg
            st["share"] = eng.shared_block0_buffer(len(st["frames"])) if (not st["concurrent"] and not args.no_share_block0) else None
            if st["concurrent"] and "ws2" not in st:
                st["ws2"] = eng.second_workspace(len(st["frames"]))
                st["side"] = torch.cuda.Stream(device=dev)
                torch.cuda.synchronize()
            with torch.cuda.graph(g):
                if st["concurrent"]:
                    x_in = xin(st)
                    main = torch.cuda.current_stream()
                    st["side"].wait_stream(main)                      # fork: the uncond branch on a second captured stream
                    kc, vc, ck, cv, crows = caches[0]
                    eng.forward(x_in, st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=st["fc"], cross_rows=crows, attn_history=st["hist"][0])
                    with torch.cuda.stream(st["side"]):
                        kc, vc, ck, cv, crows = caches[1]
                        eng.forward(x_in, st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=st["fu"], cross_rows=crows, workspace=st["ws2"],
                                    attn_history=st["hist"][1])
                    main.wait_stream(st["side"])                      # join
                else:
                    for which, out in ((0, st["fc"]), (1, st["fu"])):
                        kc, vc, ck, cv, crows = caches[which]
                        eng.forward(xin(st), st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, out=out, cross_rows=crows,
                                    share_out=st["share"] if which == 0 else None, share_in=st["share"] if which == 1 else None,
                                    attn_history=st["hist"][which])
                sched.step_cfg_table(st["fc"], st["fu"], st["lat"], st["t"])
            st["graph"], st["replays"] = g, 0
    concurrent_stages = [i for i, st in enumerate(stage_state) if st.get("concurrent")]
    stage_state_share = [st.get("share") is not None for st in stage_state]      # (stage_state is cleared before the VAE leg)
    if pair is not None and not args.eager:
        # CFG pair: each rank's forward is a hipGraph (the per-step exchange of the two flow predictions is issued by the host)
        for st in stage_state:
            kc, vc, ck, cv, crows = caches[0]
            st["t"].fill_(999.0)
            st["fwd_graph"] = eng.capture(xin(st), st["t"], st["frames"], st["ws"], st["vis"], kc, vc, ck, cv, st["mine"], cross_rows=crows,
                                          attn_history=st["hist"][0])
    # same-run calibration of THIS box: what it sustains on nothing but MFMAs (random operands), per instruction shape, right before
    # the warm-up steps (which bring clocks / power back to the workload's own steady state before the timed region)
    probe = None
    if args.probe_seconds > 0:
        probe = {}
        torch.cuda.synchronize()
        for shape, name in ((32, "v_mfma_f32_32x32x16_bf16"), (16, "v_mfma_f32_16x16x32_bf16")):
            tf = C.c_double(0.0)
            _lib.check(lib.mmpl_probe_mfma_tflops(shape, args.probe_seconds, C.byref(tf)), "mmpl_probe_mfma_tflops")
            probe[name] = tf.value
    for i in range(args.warmup):
        one_step(i)
    handoff_exchange(stage_state[1])       # untimed: the p2p communicators exist before the timed region whatever --warmup is
    if pair is not None:
        pair.exchange(stage_state[0]["mine"], stage_state[0]["flow"])
    torch.cuda.current_stream().wait_stream(side)
    barrier()
    if not args.no_profile and not graphed:
        lib.mmpl_profile_enable(1 if args.profile_all else ((1 << 1) << 1))      # default: kind 1 = self-attention only
    # the K timed steps keep rotating through the four stage shapes; every step is also bracketed by a HIP event pair so
    # that the chunk time can be assembled per stage (exact for any K, not only multiples of 4)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if attn_stats is not None:
        attn_stats.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        one_step(args.warmup + i)
        ev[i][1].record()
    torch.cuda.current_stream().wait_stream(side)
    barrier()
    elapsed = time.perf_counter() - t0
    step_s = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in ev]
    attn_blocks, attn_redone, attn_waves_bad, attn_predicted, attn_remembered = eng.read_attn_stats() if attn_stats is not None else (None,) * 5   # of the timed region (graph replays included)
    # ---- eager pass (after the timed region when that one replayed graphs): one rotation of the four stages with a hipEvent
    # pair around every self-attention launch (all kernel classes with --profile-all) -> `eager` figures and `roofline`
    eager_step_s = None
    if graphed:
        for st in stage_state:
            st["sched"].set_timesteps(50, shift=5.0)
        for i in range(4):
            one_step(i, eager=True)                         # untimed: eager launch path warm
        torch.cuda.synchronize()
        if not args.no_profile:
            lib.mmpl_profile_enable(1 if args.profile_all else ((1 << 1) << 1))
        n_eager = 8
        eev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_eager)]
        for i in range(n_eager):
            eev[i][0].record()
            one_step(i, eager=True)
            eev[i][1].record()
        torch.cuda.synchronize()
        eager_step_s = [e0.elapsed_time(e1) * 1e-3 for e0, e1 in eev]
    prof = None
    if not args.no_profile:
        n = len(KIND_NAMES)
        ms, fl, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        lib.mmpl_profile_read(n, ms, fl, cnt)
        lib.mmpl_profile_enable(0)
        prof = {KIND_NAMES[k]: dict(ms=ms[k], flops=fl[k], launches=int(cnt[k])) for k in range(n) if cnt[k]}
    vae_s = None
    if rank == 0 and not args.no_vae:
        # Wan 3D-VAE decode of one 21-latent-frame chunk (reported separately from the DiT metric)
        try:
            from mmpl_amd.synthetic import vae_state_dict
            from mmpl_amd.vae import VaeEngine
            from mmpl_amd.wan_wrapper import WanVAEWrapper
            for st in stage_state:
                st.clear()
            ve = VaeEngine(lat_h, lat_w, dev)
            ve.load_state_dict(vae_state_dict(seed=7))
            z = torch.randn(21, 16, lat_h, lat_w, device=dev).to(torch.bfloat16)
            ve.decode(z[:2], WanVAEWrapper.mean, WanVAEWrapper.std)
            torch.cuda.synchronize()
            tv = time.perf_counter()
            ve.decode(z, WanVAEWrapper.mean, WanVAEWrapper.std)
            torch.cuda.synchronize()
            vae_s = time.perf_counter() - tv
        except Exception as e:
            vae_s = f"failed: {e!r}"
    # per-stage mean step time; a stage the K steps did not reach is priced at the measured FLOP rate of the others
    stage_flops = [dit_forward_flops(cfg, S, q, kv) for q, kv in stage_shapes]
    from mmpl_amd.stage_plan import assemble_chunk_seconds
    stage_s, chunk_s = assemble_chunk_seconds(step_s, args.warmup, stage_flops)     # 4 stages x (50 denoise steps + 1 refresh pair)
    if dist is not None:
        tt = torch.tensor([elapsed, chunk_s], device="cpu" if gloo else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, chunk_s = tt[0].item(), tt[1].item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        # One video: chunk c+1 starts when chunk c's anchor stage is done, and chunks >= 2 consist of that stage and the two
        # in-fill stages (their first two frames come from the hand-off), so the wavefront keeps at most
        # (s1 + s2 + s3) / s1 lanes busy (measured stage times of THIS run).
        occupancy = (stage_s[1] + stage_s[2] + stage_s[3]) / stage_s[1]
        lanes_busy = min(float(n_lanes), occupancy)
        value = lanes_busy * 21.0 / chunk_s
        chunk_flops = 102.0 * sum(stage_flops)
        achieved_pf = (2.0 if pair is None else 1.0) * 51.0 * sum(stage_flops) / chunk_s / 1e15    # forwards per rank-step
        # ... and the FLOPs of the launches that are actually in the timed steps: the text cross-attention over the collapsed key set,
        # and block 0's self-attention + output projection once per step where the branches run back to back (share_out / share_in)
        from mmpl_amd.stage_plan import dit_forward_flops_executed
        exec_step = []
        for si, (q, kv) in enumerate(stage_shapes):
            st_share = use_graph and stage_state_share[si]
            T_txt = cfg.get("text_len", 512)
            keys = lambda rows: rows + 1 if 0 <= rows <= T_txt - 2 else T_txt          # mmpl_dit_forward's cross_rows rule
            per_branch = [dit_forward_flops_executed(cfg, S, q, kv, keys(caches[b][4]), block0_self_attn_shared=(b == 1 and st_share))
                          for b in range(len(caches))]
            exec_step.append(sum(per_branch))
        executed_pf = 51.0 * sum(exec_step) / chunk_s / 1e15
        res = {
            "metric": "video_latent_frames_per_sec", "value": value, "unit": "latent-frames/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1,        # the ranks RCCL actually connected
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            **({"value_is_modelled": "--rotation with N > 1: `value` = lanes the wavefront can keep busy x 21 / chunk time from this run's stage "
                                     "times -- an occupancy model; the default N > 1 line measures the dependent wavefront instead"} if world > 1 else {}),
            **({"functional_only": "gloo backend: ranks share GPUs, exchanges staged through the host -- exercises the N > 1 code "
                                   "paths, not a scaling measurement"} if gloo else {}),
            "config": {"workload": f"Wan2.1-{args.mode.upper()}-{args.model} {args.res} chunk-AR denoise step (cond+uncond DiT forward, CFG, UniPC), "
                                   f"rotating the four {args.mode.upper()} denoise stages {stage_shapes} (query, attended frames); one "
                                   f"21-latent-frame chunk per GPU = 204 step-equivalents",
                       "frame_seqlen": S, "latent_hw": [lat_h, lat_w], "sampling_steps": 50, "guidance_scale": 5.0,
                       "timed_path": ("one hipGraph replay per denoise step (2 DiT forwards" +
                                      (f", as two parallel graph branches in stages {concurrent_stages}" if concurrent_stages else "") +
                                      ("" if args.no_share_block0 else "; block 0's self-attention computed once per step where the branches run back to back") +
                                      " + fused CFG/UniPC, device-resident step tables)"
                                      if use_graph else ("one hipGraph replay per forward + host-issued flow exchange + fused CFG/UniPC launch"
                                                         if pair is not None and not args.eager else "eager launches")),
                       "model_type": "i2v (in_dim 36 + CLIP image stream)" if args.i2v_model else "t2v",
                       "parallelism": (f"chunk-per-rank x{world}" if pair is None else f"{n_lanes} chunk lanes x 2 (cond|uncond) CFG split, "
                                       "per-step 2-rank all-gather of flow predictions") +
                                      (" + RCCL p2p anchor hand-off lane->lane+1" if world > 1 else "") +
                                      (f"; one video keeps min(lanes, (s1+s2+s3)/s1 = {occupancy:.2f}) = {lanes_busy:.2f} lanes busy" if world > 1 else "")},
            "value_independent_chunks": n_lanes * 21.0 / chunk_s,
            "sec_per_denoise_step": elapsed / args.steps,
            "sec_per_denoise_step_by_stage": stage_s,
            "sec_per_chunk_extrapolated": chunk_s,
            "achieved_pflops_per_gpu": achieved_pf,
            "mfma_frac_whole_step": achieved_pf * 1e3 / MFMA_PEAK_TFLOPS,
            "executed_pflops_per_gpu": executed_pf,
            "mfma_frac_executed": executed_pf * 1e3 / MFMA_PEAK_TFLOPS,
            "flop_accounting": "achieved_* = the reference's algorithmic FLOPs (SURVEY.md 8d: both CFG branches in full, 512 cross-attention keys) / time; "
                               "executed_* = the FLOPs of the launches in the timed step graphs (cross-attention over the collapsed key set; block 0's "
                               "self-attention + o-projection once per step where share_out / share_in applies).  The chunk model prices the refresh "
                               "pair of a stage as one more step (the pipeline replays the per-forward graphs there, without the share: < 0.02 % of a chunk)",
            "vae_decode_s_per_chunk": vae_s,
            # data dependence of the self-attention kernel IN THE TIMED REGION: 256-row query blocks whose max-free FAST softmax pass
            # overflowed / underflowed and were redone by the GENERAL pass (attn_w64.hip); caches hold K / V written by real forwards
            **({"attn_blocks": attn_blocks, "attn_blocks_redone": attn_redone,
                "attn_blocks_redone_fraction": (attn_redone / attn_blocks) if attn_blocks else None,
                # per WAVE (64 of a block's 256 query rows): how many held a failing row themselves -- what a finer redo unit would pay for
                "attn_waves_failed_fraction": (attn_waves_bad / (4.0 * attn_blocks)) if attn_blocks else None,
                # blocks their history (an earlier failure even on remembered references) sent straight to the GENERAL pass: 1.66 FAST-pass
                # times instead of 2.66 -- and blocks whose FAST pass HELD on the lane references their history remembered: 1.0
                "attn_blocks_predicted": attn_predicted,
                "attn_blocks_predicted_fraction": (attn_predicted / attn_blocks) if attn_blocks else None,
                "attn_blocks_fast_on_remembered_reference": attn_remembered,
                "attn_blocks_fast_on_remembered_reference_fraction": (attn_remembered / attn_blocks) if attn_blocks else None} if attn_stats is not None else {}),
            "attn_history": "off (--no-attn-history): stateless self-attention" if args.no_attn_history else
                            "on: a state byte + 128 lane references per (stage, branch, layer, head, 256-row query block) carried from step to step (include/mmpl_hip.h)",
            **({"weights": f"heavy-tailed synthetic (QK-norm gains x{args.heavy_tail_gain:g}" +
                           (f" on {args.heavy_tail_heads:g} of every layer's heads" if args.heavy_tail_heads is not None else "") +
                           ", six massive-activation channels): NOT the headline workload"}
               if args.heavy_tail else {}),
        }
        if eager_step_s is not None:
            e_stage, e_chunk = assemble_chunk_seconds(eager_step_s, 0, stage_flops)
            res["eager"] = {"sec_per_denoise_step_by_stage": e_stage, "sec_per_chunk_extrapolated": e_chunk,
                            "latent_frames_per_sec": 21.0 / e_chunk, "steps": len(eager_step_s),
                            "note": "same steps as plain launches (per-kernel hipEvent pairs on), run right after the timed graph replays"}
        if prof and "attn_self" in prof:
            a = prof["attn_self"]
            ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
            traffic, traffic_src = None, None
            if args.model == "14B" and args.res == "720p" and args.mode == "t2v" and not args.heavy_tail:
                # Fabric bytes per op cannot be collected inside this process (rocprofv3 --pmc wraps the program): the committed PMC
                # passes over THIS command (tools/r06_gpu.sh hbm: bench.py under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
                # passes, mean over the rotation's self-attention ops).  The file is named explicitly by profiles/PMC_TRAFFIC.json,
                # written by the collection script -- not picked by a sorted glob.
                idx = os.path.join(ROOT, "profiles", "PMC_TRAFFIC.json")
                if os.path.exists(idx):
                    name = json.load(open(idx))["attention_traffic_file"]
                    traffic = json.load(open(os.path.join(ROOT, "profiles", name)))["mean_hbm_bytes_per_op"]
                    traffic_src = (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py itself, "
                                   "mean of the rotation's self-attention ops; a PREVIOUSLY collected figure, not a measurement of this run)")
            res["roofline"] = {"bound": "mfma", "kernel": "attn_w64_kernel (self-attention over the KV-slot page table; one op = main launch + split-KV tail launch + merge)",
                               "measured_in": ("eager pass right after the timed graph replays (hipEvent pair per launch on the launch stream)"
                                               if graphed else "timed region (hipEvent pair per launch on the launch stream)"),
                               "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                               **({"sustained_probe_tflops": probe,
                                   "frac_of_sustained": ach / probe["v_mfma_f32_32x32x16_bf16"],
                                   "sustained_probe_note": f"mmpl_probe_mfma_tflops: {args.probe_seconds:g} s per shape of back-to-back MFMAs on random bf16 operands (one wave "
                                                           "per SIMD, accumulators in AGPRs) on this box right before the warm-up steps; the attention kernel issues the "
                                                           "32x32x16 shape: frac_of_sustained = achieved / that shape's sustained rate (the chip is power-limited: the nominal "
                                                           "`peak` is not reachable on non-zero operands)"} if probe else {}),
                               "traffic": traffic, "traffic_unit": "fabric bytes per op (L2 <-> memory side; Infinity-Cache hits are counted, so this is an upper bound of HBM bytes)",
                               "traffic_source": traffic_src,
                               "avg_launch_ms": a["ms"] / a["launches"], "launches": a["launches"],
                               "algorithmic_flops_per_launch": a["flops"] / a["launches"]}
            # share of the timed wall clock spent in the roofline kernel (per rank; exact, unlike a share of timed kernels)
            res["roofline"]["time_share_of_step"] = round(a["ms"] * 1e-3 / sum(eager_step_s if eager_step_s is not None else step_s), 4)
            if args.profile_all:
                tot = sum(v["ms"] for v in prof.values())
                res["kernel_time_share"] = {k: round(v["ms"] / tot, 4) for k, v in prof.items()}
            if "gemm" in prof:
                g = prof["gemm"]
                res["gemm_tflops"] = g["flops"] / (g["ms"] * 1e-3) / 1e12
                if probe:
                    res["gemm_frac_of_sustained"] = res["gemm_tflops"] / probe["v_mfma_f32_16x16x32_bf16"]
        if isinstance(vae_s, float):
            # hook-counted conv + attention FLOPs of the reference decoder (vae.py:545-569), 10.54 TFLOP for the first
            # latent frame + 31.77 TFLOP per further one at 720p (90 x 160 latents), proportional to h * w
            vfl = (10.54e12 + 20 * 31.77e12) * (lat_h * lat_w) / (90.0 * 160.0)
            res["vae_roofline"] = {"bound": "mfma", "kernel": "mmpl_vae_decode (21 latent -> 81 pixel frames; implicit-GEMM causal conv3d / conv2d + fused norm / upsample kernels)",
                                   "achieved": vfl / vae_s / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": vfl / vae_s / 1e12 / MFMA_PEAK_TFLOPS, "algorithmic_flops": vfl, "seconds": vae_s,
                                   "note": "reported beside the DiT metric, not part of `value` (0.35 % of a chunk)"}
        if not args.no_cpu_baseline:           # rank 0 only (this branch), whatever the world size
            try:
                res["cpu_baseline"] = cpu_baseline(cfg, lat_h, lat_w, stage_shapes, args.cpu_budget_s)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU measurement over it
                res["cpu_baseline"] = {"value": None, "unit": "latent-frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
