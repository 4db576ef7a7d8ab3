"""Frame / token geometry of the Wan2.1 latent grid (lifts the reference's literals 1560 / 32760 / [.,.,16,60,104],
MMPL_t2v/wan/modules/causal_fps_model.py:75,194,206 and pipeline/casual_fps_inference.py:82,95,461)."""
from __future__ import annotations

from dataclasses import dataclass

RESOLUTIONS = {"480p": (60, 104), "720p": (90, 160)}   # latent (h, w); pixel = 8x


@dataclass(frozen=True)
class Geometry:
    lat_h: int
    lat_w: int
    frames_per_chunk: int = 21
    n_slots: int = 15          # (32760 - 6*1560) / 1560, casual_fps_inference.py:461

    @classmethod
    def named(cls, name: str) -> "Geometry":
        h, w = RESOLUTIONS[name]
        return cls(h, w)

    @property
    def grid_h(self) -> int:
        return self.lat_h // 2

    @property
    def grid_w(self) -> int:
        return self.lat_w // 2

    @property
    def frame_seqlen(self) -> int:
        return self.grid_h * self.grid_w

    @property
    def pixel_hw(self):
        return self.lat_h * 8, self.lat_w * 8
