"""Image conditioning of the Wan-I2V model type, built the way ``WanI2V.generate`` builds it
(MMPL_t2v/wan/image2video.py:207-246): a 4-channel first-frame mask and the VAE latents of [image, 80 zero frames]
stacked into the 20-channel conditioning video ``y`` that is concatenated to the latents on the channel axis
(wan/modules/model.py:680-681), plus the CLIP ViT-H tokens ``clip_fea`` every block's cross-attention also attends to
(model.py:708-712).  Host-side orchestration only: VAE encode and the CLIP tower are the HIP engines."""
from __future__ import annotations

from typing import Dict

import torch


def first_frame_mask(n_latent_frames: int, lat_h: int, lat_w: int, device) -> torch.Tensor:
    """image2video.py:207-214: ones on pixel frame 0, zeros elsewhere, the first frame repeated 4x, folded to
    [4, n_latent_frames, h, w] (4 pixel frames per latent frame on the channel axis)."""
    T = 1 + 4 * (n_latent_frames - 1)
    msk = torch.ones(1, T, lat_h, lat_w, device=device)
    msk[:, 1:] = 0
    msk = torch.cat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
    msk = msk.view(1, msk.shape[1] // 4, 4, lat_h, lat_w)
    return msk.transpose(1, 2)[0]


def build_image_condition(vae, clip, image: torch.Tensor, n_latent_frames: int = 21) -> Dict[str, torch.Tensor]:
    """image: [3, H, W] in [-1, 1] at the pixel size of the VAE's geometry (8 lat_h x 8 lat_w).
    vae: WanVAEWrapper; clip: CLIPVisionTower.  -> {"clip_fea": [257, clip_dim] bf16, "y": [20, n_latent_frames, h, w] bf16}."""
    dev = vae.model.device
    lat_h, lat_w = vae.model.lat_h, vae.model.lat_w
    img = image.to(device=dev, dtype=torch.bfloat16)
    assert img.shape == (3, 8 * lat_h, 8 * lat_w), (tuple(img.shape), lat_h, lat_w)
    clip_fea = clip.visual([img[:, None, :, :].float()])[0]                      # image2video.py:232 (one image -> [257, dim])
    T = 1 + 4 * (n_latent_frames - 1)
    clipv = torch.zeros(3, T, 8 * lat_h, 8 * lat_w, dtype=torch.bfloat16, device=dev)
    clipv[:, 0] = img                                                            # :236-244 (the interpolate there is a resize to this size)
    lat = vae.encode_to_latent(clipv.unsqueeze(0))[0]                            # [F, 16, h, w] normalised mu
    y = torch.cat([first_frame_mask(n_latent_frames, lat_h, lat_w, dev), lat.permute(1, 0, 2, 3).float()], dim=0)
    return {"clip_fea": clip_fea.to(torch.bfloat16), "y": y.to(torch.bfloat16).contiguous()}
