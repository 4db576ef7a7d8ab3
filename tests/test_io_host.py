"""Host-side I/O of the entry scripts: prompt / image datasets and the video writer."""
import json

import numpy as np
import pytest
import torch
from PIL import Image

from mmpl_amd.utils.dataset import TextDataset, TextImagePairDataset
from mmpl_amd.utils.video_io import read_mjpeg_avi, write_video


def test_text_dataset(tmp_path):
    p = tmp_path / "prompts.txt"
    p.write_text("a cat \nA dog on grass\n", encoding="utf-8")
    e = tmp_path / "ext.txt"
    e.write_text("a cat, detailed\na dog, detailed\n", encoding="utf-8")
    ds = TextDataset(str(p))
    assert len(ds) == 2 and ds[0] == {"prompts": "a cat", "idx": 0}
    ds = TextDataset(str(p), str(e))
    assert ds[1] == {"prompts": "A dog on grass", "idx": 1, "extended_prompts": "a dog, detailed"}


def test_text_image_pair_dataset(tmp_path):
    (tmp_path / "26-15").mkdir()
    meta = []
    for i in range(3):
        Image.fromarray(np.full((15, 26, 3), 40 * i, np.uint8)).save(tmp_path / "26-15" / f"{i}.png")
        meta.append(dict(file_name=f"{i}.png", caption=f"cap {i}", target_crop=dict(target_bbox=[0, 0, 26, 15], target_ratio="26-15"),
                         type="t", origin_width=26, origin_height=15))
    (tmp_path / "target_crop_info_26-15.json").write_text(json.dumps(meta))
    ds = TextImagePairDataset(str(tmp_path), transform=lambda im: torch.from_numpy(np.asarray(im).copy()), pad_to_multiple_of=2)
    assert len(ds) == 4 and ds.pre_pad_len == 3
    it = ds[3]
    assert it["prompts"] == "cap 2" and it["origin_size"] == (26, 15) and it["image"].shape == (15, 26, 3) and int(it["image"][0, 0, 0]) == 80
    with pytest.raises(FileNotFoundError):
        TextImagePairDataset(str(tmp_path / "26-15"))


def test_write_video_roundtrip(tmp_path):
    T, H, W = 5, 48, 64
    yy, xx = np.mgrid[0:H, 0:W]
    frames = np.stack([np.stack([(xx * 3 + 10 * t) % 256, (yy * 4) % 256, np.full_like(xx, 30 * t)], -1) for t in range(T)]).astype(np.uint8)
    out = write_video(str(tmp_path / "clip.mp4"), torch.from_numpy(frames), fps=16)
    if out.endswith(".avi"):
        back = read_mjpeg_avi(out)
        assert back.shape == frames.shape
        assert np.abs(back.astype(int) - frames.astype(int)).mean() < 6.0        # JPEG q92 on a synthetic gradient
        hdr = open(out, "rb").read(64)
        assert hdr[:4] == b"RIFF" and hdr[8:12] == b"AVI "
