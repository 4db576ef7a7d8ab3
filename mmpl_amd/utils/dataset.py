"""Prompt / image datasets of the inference entry scripts (MMPL_t2v/utils/dataset.py:12-34 TextDataset, :127-215
TextImagePairDataset; used by Wan_fps_inference_1gpu.py:81-83).  Same constructor arguments and item dictionaries; the
LMDB training datasets of that file are out of scope."""
from __future__ import annotations

import json
from pathlib import Path
from typing import NamedTuple

from PIL import Image
from torch.utils.data import Dataset


def _read_lines(path):
    with open(path, encoding="utf-8") as fh:
        return [ln.rstrip() for ln in fh]


class TextDataset(Dataset):
    """One prompt per line (trailing whitespace stripped, blank lines kept); optionally a second file with one
    "extended" prompt per line, which must have the same number of lines.  Items: {"prompts", "idx"[, "extended_prompts"]}."""

    def __init__(self, prompt_path, extended_prompt_path=None):
        prompts = _read_lines(prompt_path)
        extended = _read_lines(extended_prompt_path) if extended_prompt_path is not None else None
        if extended is not None and len(extended) != len(prompts):
            raise AssertionError(f"{extended_prompt_path}: {len(extended)} lines, {prompt_path}: {len(prompts)}")
        self._rows = [(p, None if extended is None else extended[i]) for i, p in enumerate(prompts)]
        self._has_extended = extended is not None

    @property
    def prompt_list(self):
        return [p for p, _ in self._rows]

    @property
    def extended_prompt_list(self):
        return [e for _, e in self._rows] if self._has_extended else None

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, idx):
        prompt, ext = self._rows[idx]
        item = {"prompts": prompt, "idx": idx}
        if self._has_extended:
            item["extended_prompts"] = ext
        return item


def _find_manifest(root: Path) -> Path:
    """The single ``target_crop_info_<ratio>.json`` of an image-prompt folder."""
    found = sorted(root.glob("target_crop_info_*.json"))
    if len(found) != 1:
        kind = FileNotFoundError if not found else ValueError
        raise kind(f"{root}: expected exactly one target_crop_info_*.json manifest, found {len(found)}")
    return found[0]


class _ImageRecord(NamedTuple):
    path: Path
    caption: str
    bbox: list
    ratio: str
    kind: str
    size: tuple


class TextImagePairDataset(Dataset):
    """Image + caption pairs for the I2V entry scripts (interface of MMPL_t2v/utils/dataset.py:127-215: same constructor
    arguments and item keys).  Folder layout: ``<data_dir>/target_crop_info_<ratio>.json`` -- a JSON list of
    {file_name, caption, target_crop: {target_bbox, target_ratio}, type, origin_width, origin_height} -- next to the image
    folder ``<data_dir>/<ratio>/``.  Every manifest row is resolved to a record up front (a missing image fails at
    construction, not at item time); ``eval_first_n`` keeps a prefix, ``pad_to_multiple_of`` repeats the last record so that
    the length divides evenly across ranks (``pre_pad_len`` is the un-padded length)."""

    def __init__(self, data_dir, transform=None, eval_first_n=-1, pad_to_multiple_of=None):
        root = Path(data_dir)
        manifest = _find_manifest(root)
        ratio = manifest.stem[len("target_crop_info_"):]
        folder = root / ratio
        if not folder.is_dir():
            raise FileNotFoundError(f"{folder}: image folder named by {manifest.name} is missing")
        rows = json.loads(manifest.read_text())
        if eval_first_n >= 0:
            rows = rows[:eval_first_n]
        records = [_ImageRecord(folder / r["file_name"], r["caption"], r["target_crop"]["target_bbox"],
                                r["target_crop"]["target_ratio"], r["type"], (r["origin_width"], r["origin_height"])) for r in rows]
        absent = [str(rec.path) for rec in records if not rec.path.is_file()]
        if absent:
            raise FileNotFoundError(f"{len(absent)} image(s) listed in {manifest.name} do not exist, first: {absent[0]}")
        self.transform = transform
        self.image_dir = folder
        self.pre_pad_len = len(records)
        shortfall = -len(records) % pad_to_multiple_of if pad_to_multiple_of else 0
        self._records = records + records[-1:] * shortfall

    def __len__(self):
        return len(self._records)

    def __getitem__(self, idx):
        rec = self._records[idx]
        with Image.open(rec.path) as fh:
            image = fh.convert("RGB")
        if self.transform is not None:
            image = self.transform(image)
        return dict(image=image, prompts=rec.caption, target_bbox=rec.bbox, target_ratio=rec.ratio, type=rec.kind,
                    origin_size=rec.size, idx=idx)
