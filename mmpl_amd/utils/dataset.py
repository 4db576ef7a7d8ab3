"""Prompt / image datasets of the inference entry scripts (MMPL_t2v/utils/dataset.py:12-34 TextDataset, :127-215
TextImagePairDataset; used by Wan_fps_inference_1gpu.py:81-83).  Same constructor arguments and item dictionaries; the
LMDB training datasets of that file are out of scope."""
from __future__ import annotations

import json
from pathlib import Path

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


class TextImagePairDataset(Dataset):
    """data_dir holds ONE ``target_crop_info_<ratio>.json`` (list of {file_name, caption, target_crop{target_bbox,
    target_ratio}, type, origin_width, origin_height}) and the image folder ``<ratio>/``."""

    def __init__(self, data_dir, transform=None, eval_first_n=-1, pad_to_multiple_of=None):
        self.transform = transform
        data_dir = Path(data_dir)
        metadata_files = list(data_dir.glob("target_crop_info_*.json"))
        if not metadata_files:
            raise FileNotFoundError(f"No metadata file found in {data_dir}")
        if len(metadata_files) > 1:
            raise ValueError(f"Multiple metadata files found in {data_dir}")
        metadata_path = metadata_files[0]
        self.image_dir = data_dir / metadata_path.stem.split("_")[-1]
        if not self.image_dir.exists():
            raise FileNotFoundError(f"Image directory not found: {self.image_dir}")
        with open(metadata_path, "r") as f:
            self.metadata = json.load(f)
        if eval_first_n != -1:
            self.metadata = self.metadata[:eval_first_n]
        for item in self.metadata:
            if not (self.image_dir / item["file_name"]).exists():
                raise FileNotFoundError(f"Image not found: {self.image_dir / item['file_name']}")
        self.dummy_prompt = "DUMMY PROMPT"
        self.pre_pad_len = len(self.metadata)
        if pad_to_multiple_of is not None and len(self.metadata) % pad_to_multiple_of != 0:
            self.metadata += [self.metadata[-1]] * (pad_to_multiple_of - len(self.metadata) % pad_to_multiple_of)

    def __len__(self):
        return len(self.metadata)

    def __getitem__(self, idx):
        item = self.metadata[idx]
        image = Image.open(self.image_dir / item["file_name"]).convert("RGB")
        if self.transform:
            image = self.transform(image)
        return {"image": image, "prompts": item["caption"], "target_bbox": item["target_crop"]["target_bbox"],
                "target_ratio": item["target_crop"]["target_ratio"], "type": item["type"],
                "origin_size": (item["origin_width"], item["origin_height"]), "idx": idx}
