"""The C-ABI library loads on a CPU-only box and exports every symbol include/mmpl_hip.h declares (no compute)."""
import ctypes
import os
import re

from mmpl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mmpl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mmpl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmpl_hip.h but not exported by libmmpl_hip.so"
    assert sorted(_lib.SYMBOLS) == names, (sorted(set(names) ^ set(_lib.SYMBOLS)))


def test_version_and_error_strings():
    lib = _lib.load()
    assert b"gfx950" in lib.mmpl_version()
    assert isinstance(lib.mmpl_last_error(), bytes)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.MmplDitConfig) == 13 * 4
    assert ctypes.sizeof(_lib.MmplUniPCStep) == 15 * 4


def test_weight_slot_names():
    lib = _lib.load()
    cfg = _lib.MmplDitConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, freq_dim=256, in_dim=16, out_dim=16,
                             text_len=512, eps=1e-6, lat_h=16, lat_w=24, max_frames=7)
    assert lib.mmpl_dit_num_weights(ctypes.byref(cfg)) == 16 + 2 * 22
    assert lib.mmpl_dit_weight_name(0, 0) == b"patch_embedding.weight"
    assert lib.mmpl_dit_weight_name(21, 1) == b"ffn.2.bias"
    assert lib.mmpl_dit_weight_name(22, 1) is None


def test_no_oracle_import_in_product():
    """the product path must never route through the oracle (or any CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "mmpl_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
