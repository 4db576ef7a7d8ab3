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
    args = ap.parse_args(argv)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # ChunkHandoff.recv / CfgPair.broadcast park a rank until the previous lane has finished its anchor stage: minutes per
        # lane at 14B/720p, far beyond the 10-minute default watchdog once there are more than a few lanes
        dist.init_process_group("nccl", timeout=datetime.timedelta(hours=12))
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    torch.set_grad_enabled(False)

    from .pipeline import CausalFPSInferencePipeline
    from .wan_wrapper import SyntheticTextEncoder, WanFPSWrapper, WanVAEWrapper
    config = load_config(args.config_path)
    config.sampling_steps = args.sampling_steps
    geo = Geometry(*args.latent_hw) if args.latent_hw else Geometry.named(args.resolution)
    mcfg = WAN_CONFIGS[args.model]
    gen = WanFPSWrapper(**config.model_kwargs, is_causal=True, model_config=mcfg if args.synthetic else None, geometry=geo, device=dev)
    if args.synthetic:
        gen.load_state_dict(dit_state_dict(mcfg, seed=1, device=dev))
        enc = SyntheticTextEncoder(mcfg.get("text_dim", 4096), dev)
        vae = WanVAEWrapper(geometry=geo, device=dev, state_dict=vae_state_dict(seed=2))
    else:
        enc, vae = None, None       # reference checkpoints under ../wan_models (wan_wrapper.local_wan_path)
    mode = "i2v" if args.i2v else "t2v"
    pipe = CausalFPSInferencePipeline(config, dev, generator=gen, text_encoder=enc, vae=vae, device_cond=dev, device_uncond=dev, save=None,
                                      mode=mode, geometry=geo)
    image_latent = None
    if args.i2v:
        # I2V/Wan_fps_inference_1gpu.py:74-79,100-104: Resize -> ToTensor -> Normalize(0.5, 0.5) -> vae.encode_to_latent
        import numpy as np
        from PIL import Image
        H, W = geo.pixel_hw
        img = Image.open(args.image).convert("RGB").resize((W, H), Image.BILINEAR)
        px = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float() / 255.0
        px = ((px - 0.5) / 0.5).unsqueeze(0).unsqueeze(2).to(device=dev, dtype=torch.bfloat16)      # [1, 3, 1, H, W]
        image_latent = pipe.vae.encode_to_latent(px).to(torch.bfloat16)                                # [1, 1, 16, h, w]
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
    shape = [1, args.num_output_frames, 16, geo.lat_h, geo.lat_w]
    for idx, prompt in enumerate(prompts):
        # noise for every chunk is drawn in order from one seeded generator on every rank (the reference draws it on the
        # main thread in chunk order, ..._parallel_4gpu_20s.py:170,232-251)
        g = torch.Generator(device="cpu").manual_seed(args.seed)
        noises = [torch.randn(shape, generator=g).to(torch.bfloat16) for _ in range(args.duration)]
        if world == 1:
            videos, initial = [], image_latent
            for c in range(args.duration):
                video, _ = pipe.inference(noises[c].to(dev), [prompt], initial_latent=initial, return_latents=True)
                initial = rolling_initial_latent(pipe.vae, video)
                videos.append(video.cpu())
        else:
            pair, heads, lay = (CfgPair.build(world, dev, True) if args.cfg_split else (None, None, None))
            pipe.cfg_pair = pair
            rank = dist.get_rank()
            ho = (ChunkHandoff((1, 3 if args.i2v else 8, 16, geo.lat_h, geo.lat_w), dev, group=heads)
                  if pair is None or pair.role == 0 else None)

            def make_chunk(c, initial, sink):
                pipe.handoff_sink = sink
                if c == 0:
                    initial = image_latent
                video, _ = pipe.inference(noises[c].to(dev), [prompt], initial_latent=initial, return_latents=True)
                return video

            videos = run_chunk_wavefront(make_chunk, args.duration, ho, lambda t: handoff_to_initial_latent(pipe.vae, t.to(dev)),
                                         pair=pair, lane=lay["lane_of"][rank] if lay else rank, n_lanes=world // 2 if lay else world,
                                         initial_like=torch.empty([1, 2, 16, geo.lat_h, geo.lat_w], device=dev, dtype=torch.bfloat16))
        if videos is not None:
            full = stitch_chunks(videos)                                     # [1, T, 3, H, W] in [0, 1]
            out = (full[0].permute(0, 2, 3, 1) * 255.0).clamp(0, 255).to(torch.uint8)
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
