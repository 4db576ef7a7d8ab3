"""Entry point with the reference's CLI (MMPL_t2v/Wan_fps_inference_1gpu.py:21-36 and the `_parallel_4gpu_*` scripts).

    python -m mmpl_amd.cli --config_path configs/self_forcing_df.yaml --checkpoint_path t2v_14B_8k.pt \
           --data_path prompts.txt --output_folder out --duration 2 --seed 0
    torchrun --nproc-per-node 4 --master-addr 127.0.0.1 -m mmpl_amd.cli ... --duration 4     # chunk wavefront, RCCL hand-off

Single process = the reference's sequential rollout (chunk k+1 starts from the last 5 pixel frames of chunk k,
`Wan_fps_inference_1gpu.py:164-203`); several processes = the parallel scripts' wavefront (chunk c on rank c mod W,
anchors handed over right after the anchor stage).  `--synthetic` runs without checkpoints (seeded weights and text
embeddings) -- the only mode that can run in this repo's test environments.  Output: per prompt a uint8 tensor
`[T, H, W, 3]` saved with torch.save (mp4 muxing via torchvision is outside the hot path; fps = 16).
"""
from __future__ import annotations

import argparse
import os
import types

import torch
import yaml

from .geometry import Geometry
from .handoff import (CfgPair, ChunkHandoff, handoff_to_initial_latent, rolling_initial_latent, run_chunk_wavefront,
                      stitch_chunks)
from .synthetic import WAN_CONFIGS, dit_state_dict, vae_state_dict

DEFAULTS = dict(num_train_timestep=1000, timestep_shift=5.0, guidance_scale=5.0, independent_first_frame=False,
                negative_prompt="", model_kwargs={"timestep_shift": 5.0})


def load_config(path):
    cfg = dict(DEFAULTS)
    if path:
        with open(path) as f:
            cfg.update(yaml.safe_load(f) or {})
    return types.SimpleNamespace(**cfg)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config_path", type=str)
    ap.add_argument("--checkpoint_path", type=str)
    ap.add_argument("--data_path", type=str)
    ap.add_argument("--output_folder", type=str, default="outputs")
    ap.add_argument("--num_output_frames", type=int, default=21)
    ap.add_argument("--use_ema", action="store_true")
    ap.add_argument("--i2v", action="store_true", help="image-to-video: the VAE-encoded image is latent frame 0 (MMPL_i2v)")
    ap.add_argument("--image", type=str, help="input image for --i2v (any PIL-readable file)")
    ap.add_argument("--i2v_model", action="store_true",
                    help="with --i2v: the Wan-I2V MODEL TYPE (in_dim 36, CLIP ViT-H image cross-attention in every block; Wan2.1-I2V-14B, "
                         "BASELINE configs[4]) instead of the T2V backbone MMPL's own I2V scripts run; the conditioning video y and the CLIP "
                         "features are rebuilt per chunk from the chunk's first pixel frame (wan/image2video.py:207-246)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--duration", type=int, default=3, help="number of 21-latent-frame chunks")
    ap.add_argument("--resolution", default="480p", choices=["480p", "720p"])
    ap.add_argument("--model", default="14B", choices=list(WAN_CONFIGS))
    ap.add_argument("--synthetic", action="store_true", help="seeded synthetic weights / text embeddings (no checkpoints needed)")
    ap.add_argument("--sampling_steps", type=int, default=50)
    ap.add_argument("--latent_hw", type=int, nargs=2, default=None, help="override the latent size (tests)")
    ap.add_argument("--cfg_split", action="store_true",
                    help="multi-GPU: WORLD/2 chunk lanes x (cond, uncond) rank pairs -- the reference's device_cond/device_uncond "
                         "seam -- instead of WORLD chunk lanes")
    ap.add_argument("--dist_backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo stages the hand-off through the host and lets several ranks "
                         "share one GPU (tests on 1-GPU boxes)")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dist_backend == "gloo":
        local_rank %= max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # ChunkHandoff.recv / CfgPair.broadcast park a rank until the previous lane has finished its anchor stage: minutes per
        # lane at 14B/720p, far beyond the 10-minute default watchdog once there are more than a few lanes
        dist.init_process_group(args.dist_backend, timeout=datetime.timedelta(hours=12))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    torch.set_grad_enabled(False)

    from .pipeline import CausalFPSInferencePipeline
    from .wan_wrapper import SyntheticTextEncoder, WanFPSWrapper, WanVAEWrapper
    config = load_config(args.config_path)
    config.sampling_steps = args.sampling_steps
    geo = Geometry(*args.latent_hw) if args.latent_hw else Geometry.named(args.resolution)
    mcfg = WAN_CONFIGS[args.model]
    clip = None
    if args.i2v_model:
        if not (args.i2v and args.image):
            ap.error("--i2v_model needs --i2v --image")
        from .i2v_clip import CLIPVisionTower
        mcfg = dict(mcfg, model_type="i2v")
        tiny_clip = args.synthetic and args.model in ("tiny", "small")      # tests: 2 blocks instead of ViT-H's 32
        clip = CLIPVisionTower(num_layers=3 if tiny_clip else 32, device=dev)
    gen = WanFPSWrapper("Wan2.1-I2V-14B-720P" if args.i2v_model else "Wan2.1-T2V-14B", **config.model_kwargs, is_causal=True,
                        model_config=mcfg if args.synthetic else None, geometry=geo, device=dev)
    if args.i2v_model and gen.model_type != "i2v":
        raise SystemExit("--i2v_model: the checkpoint directory does not hold a model_type 'i2v' config")
    if args.synthetic:
        if args.i2v_model:
            from .synthetic import clip_visual_state_dict, dit_i2v_state_dict
            gen.load_state_dict(dit_i2v_state_dict(mcfg, seed=1, device=dev))
            clip.load_state_dict(clip_visual_state_dict(1280, 16, clip.num_layers, seed=3, device=dev))
        else:
            gen.load_state_dict(dit_state_dict(mcfg, seed=1, device=dev))
        enc = SyntheticTextEncoder(mcfg.get("text_dim", 4096), dev)
        vae = WanVAEWrapper(geometry=geo, device=dev, state_dict=vae_state_dict(seed=2))
    else:
        enc, vae = None, None       # reference checkpoints under ../wan_models (wan_wrapper.local_wan_path)
        if args.i2v_model:
            from .checkpoints import read_clip_visual
            from .wan_wrapper import local_wan_path
            clip.load_state_dict(read_clip_visual(f"{local_wan_path}/Wan2.1-I2V-14B-720P/models_clip_open-clip-xlm-roberta-large-vit-huge-14.pth"))
    mode = "i2v" if args.i2v else "t2v"
    pipe = CausalFPSInferencePipeline(config, dev, generator=gen, text_encoder=enc, vae=vae, device_cond=dev, device_uncond=dev, save=None,
                                      mode=mode, geometry=geo)
    image_latent, first_px = None, None
    if args.i2v:
        # I2V/Wan_fps_inference_1gpu.py:74-79,100-104: Resize -> ToTensor -> Normalize(0.5, 0.5) -> vae.encode_to_latent
        import numpy as np
        from PIL import Image
        H, W = geo.pixel_hw
        img = Image.open(args.image).convert("RGB").resize((W, H), Image.BILINEAR)
        px = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0
        px = ((px - 0.5) / 0.5).unsqueeze(0).unsqueeze(2).to(device=dev, dtype=torch.bfloat16)      # [1, 3, 1, H, W]
        image_latent = pipe.vae.encode_to_latent(px).to(torch.bfloat16)                                # [1, 1, 16, h, w]
        first_px = px[0, :, 0]

    def image_cond(frame):
        """the Wan-I2V model type's conditioning for a chunk whose first pixel frame is `frame` ([3, H, W] in [-1, 1])"""
        if clip is None:
            return None
        if frame is None:
            # the uncond rank of a CFG pair has no frame: pipeline.inference overwrites both tensors with the cond rank's broadcast,
            # so only the shapes matter -- no CLIP tower, no 81-frame VAE encode of a zero clip on the rank that paces the pair
            return {"clip_fea": torch.empty(257, 1280, device=dev, dtype=torch.bfloat16),
                    "y": torch.empty(20, args.num_output_frames, geo.lat_h, geo.lat_w, device=dev, dtype=torch.bfloat16)}
        from .i2v_condition import build_image_condition
        return build_image_condition(pipe.vae, clip, frame, args.num_output_frames)
    if args.checkpoint_path:
        from .checkpoints import read_mmpl_checkpoint
        pipe.generator_cond.load_state_dict(read_mmpl_checkpoint(args.checkpoint_path, use_ema=args.use_ema))

    if args.data_path:
        from .utils.dataset import TextDataset
        ds = TextDataset(args.data_path)
        prompts = [ds[i]["prompts"] for i in range(len(ds))]
    else:
        prompts = ["a cat running on the grass"]
    os.makedirs(args.output_folder, exist_ok=True)

    def to_u8(video):
        """[1, T, 3, H, W] in [0, 1] -> uint8 frames (Wan_fps_inference_1gpu.py:209-225 scales by 255 before write_video)"""
        return (video * 255.0).clamp(0, 255).to(torch.uint8)

    shape = [1, args.num_output_frames, 16, geo.lat_h, geo.lat_w]
    for idx, prompt in enumerate(prompts):
        # noise for every chunk is drawn in order from one seeded generator on every rank (the reference draws it on the
        # main thread in chunk order, ..._parallel_4gpu_20s.py:170,232-251)
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        noises = [torch.randn(shape, generator=g).to(torch.bfloat16) for _ in range(args.duration)]
        if world == 1:
            videos, initial, frame0 = [], image_latent, first_px
            for c in range(args.duration):
                video, _ = pipe.inference(noises[c].to(dev), [prompt], initial_latent=initial, return_latents=True,
                                          image_condition=image_cond(frame0))
                initial = rolling_initial_latent(pipe.vae, video)
                frame0 = video[0, -5].to(torch.bfloat16) * 2.0 - 1.0       # the next chunk starts at this chunk's 5th-last frame
                videos.append(to_u8(video))
        else:
            pair, heads, lay = (CfgPair.build(world, dev, True) if args.cfg_split else (None, None, None))
            pipe.cfg_pair = pair
            rank = dist.get_rank()
            ho = (ChunkHandoff((1, 3 if args.i2v else 8, 16, geo.lat_h, geo.lat_w), dev, group=heads)
                  if pair is None or pair.role == 0 else None)

            frame_of = {0: first_px}

            def to_initial(c_recv):
                init, frame = handoff_to_initial_latent(pipe.vae, c_recv.to(dev), return_first_frame=True)
                frame_of["next"] = frame
                return init

            def make_chunk(c, initial, sink):
                pipe.handoff_sink = sink
                pipe.handoff_poll = ho.poll if ho is not None else None
                if c == 0:
                    initial = image_latent
                video, _ = pipe.inference(noises[c].to(dev), [prompt], initial_latent=initial, return_latents=True,
                                          image_condition=image_cond(frame_of[0] if c == 0 else frame_of.get("next")))
                return to_u8(video)         # quantised on the owner: the device all-gather moves 224 MB per 720p chunk, not 896

            videos = run_chunk_wavefront(make_chunk, args.duration, ho, to_initial,
                                         pair=pair, lane=lay["lane_of"][rank] if lay else rank, n_lanes=world // 2 if lay else world,
                                         initial_like=torch.empty([1, 2, 16, geo.lat_h, geo.lat_w], device=dev, dtype=torch.bfloat16))
        if videos is not None:
            full = stitch_chunks(videos)                                     # [1, T, 3, H, W] uint8
            out = full[0].permute(0, 2, 3, 1).contiguous().cpu()
            path = os.path.join(args.output_folder, f"{idx}-0.pt")
            torch.save(out, path)
            from .utils.video_io import write_video
            vpath = write_video(os.path.join(args.output_folder, f"{idx}-0.mp4"), out, fps=16)    # Wan_fps_inference_1gpu.py:225
            print(f"[mmpl_amd.cli] prompt {idx}: {tuple(out.shape)} frames @16 fps -> {vpath} (+ {path})")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
