"""One 720p 21-latent-frame VAE decode for rocprofv3 (dev tool): python3 tools/vae_one.py [480p|720p]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmpl_amd.geometry import RESOLUTIONS  # noqa: E402
from mmpl_amd.synthetic import vae_state_dict  # noqa: E402
from mmpl_amd.vae import VaeEngine  # noqa: E402
from mmpl_amd.wan_wrapper import WanVAEWrapper  # noqa: E402

lat_h, lat_w = RESOLUTIONS[sys.argv[1] if len(sys.argv) > 1 else "720p"]
ve = VaeEngine(lat_h, lat_w, "cuda:0")
ve.load_state_dict(vae_state_dict(seed=7))
z = torch.randn(21, 16, lat_h, lat_w, device="cuda:0").to(torch.bfloat16)
ve.decode(z[:2], WanVAEWrapper.mean, WanVAEWrapper.std)
torch.cuda.synchronize()
t0 = time.perf_counter()
ve.decode(z, WanVAEWrapper.mean, WanVAEWrapper.std)
torch.cuda.synchronize()
print("decode s", time.perf_counter() - t0)
