"""``write_video(path, frames, fps=16)`` -- the reference saves results with torchvision.io.write_video
(Wan_fps_inference_1gpu.py:225).  torchvision / PyAV are used when importable (mp4/h264); otherwise (this image has
neither, nor ffmpeg) the clip is written as a Motion-JPEG AVI (every frame a baseline JPEG from PIL inside a RIFF
container with an idx1 index), which the usual players and ffmpeg read, so the rollout is still a video file."""
from __future__ import annotations

import io
import struct

import numpy as np
import torch
from PIL import Image


def _mjpeg_avi(path: str, frames: np.ndarray, fps: int, quality: int = 92) -> None:
    T, H, W, _ = frames.shape
    jpegs = []
    for t in range(T):
        buf = io.BytesIO()
        Image.fromarray(frames[t]).save(buf, format="JPEG", quality=quality)
        b = buf.getvalue()
        jpegs.append(b + (b"\x00" if len(b) & 1 else b""))               # RIFF chunks are word aligned
    max_sz = max(len(j) for j in jpegs)

    def chunk(tag: bytes, data: bytes) -> bytes:
        return tag + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")

    def lst(tag: bytes, data: bytes) -> bytes:
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data

    avih = struct.pack("<IIIIIIIIIIIIII", 1000000 // fps, max_sz * fps, 0, 0x10, T, 0, 1, max_sz, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1, fps, 0, T, max_sz, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    for j in jpegs:
        movi_body += b"00dc" + struct.pack("<I", len(j)) + j
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        off += 8 + len(j)
    body = hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body) + 4) + b"AVI " + body)


def write_video(filename: str, video_array, fps: int = 16) -> str:
    """video_array: [T, H, W, 3] uint8 (tensor or ndarray), like torchvision.io.write_video.  Returns the path written
    (``filename`` itself with torchvision; ``<stem>.avi`` for the MJPEG fallback)."""
    arr = video_array.detach().cpu().numpy() if isinstance(video_array, torch.Tensor) else np.asarray(video_array)
    assert arr.ndim == 4 and arr.shape[-1] == 3 and arr.dtype == np.uint8, "expected [T, H, W, 3] uint8"
    try:
        from torchvision.io import write_video as tv_write_video      # needs PyAV underneath
        tv_write_video(filename, torch.from_numpy(arr), fps=fps)
        return filename
    except Exception:
        out = filename.rsplit(".", 1)[0] + ".avi"
        _mjpeg_avi(out, arr, int(fps))
        return out


def read_mjpeg_avi(path: str) -> np.ndarray:
    """Decode a file written by the fallback above (tests / inspection) -> [T, H, W, 3] uint8."""
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI "
    frames, pos = [], data.index(b"movi") + 4
    end = data.index(b"idx1", pos)
    while pos < end:
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag != b"00dc":
            break
        frames.append(np.asarray(Image.open(io.BytesIO(data[pos + 8:pos + 8 + size])).convert("RGB")))
        pos += 8 + size + (size & 1)
    return np.stack(frames)
