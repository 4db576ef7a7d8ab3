"""Error behaviour of the C ABI (include/mmpl_hip.h): every entry point returns 0 / non-zero and the wrapper raises
RuntimeError(mmpl_last_error()) -- SURVEY.md 8b "errors are Python exceptions".  The argument checks run before any HIP
call, so this file needs no GPU; the one check that does reach HIP asserts that a box without a GPU fails LOUDLY
(no silent CPU fallback exists anywhere in the product path)."""
import ctypes as C

import pytest
import torch

from mmpl_amd import _lib


def _err(rc):
    assert rc != 0
    return _lib.load().mmpl_last_error().decode()


def _dit_cfg(**kw):
    base = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, freq_dim=256, in_dim=16, out_dim=16, text_len=512,
                eps=1e-6, lat_h=16, lat_w=24, max_frames=7)
    base.update(kw)
    return _lib.MmplDitConfig(**base)


def test_dit_create_argument_errors():
    lib = _lib.load()
    h = C.c_void_p()
    assert "null argument" in _err(lib.mmpl_dit_create(None, C.byref(h)))
    assert "head_dim must be 128" in _err(lib.mmpl_dit_create(C.byref(_dit_cfg(num_heads=4)), C.byref(h)))
    assert "unsupported geometry" in _err(lib.mmpl_dit_create(C.byref(_dit_cfg(lat_h=15)), C.byref(h)))
    assert "unsupported dims" in _err(lib.mmpl_dit_create(C.byref(_dit_cfg(ffn_dim=500)), C.byref(h)))
    with pytest.raises(RuntimeError, match="head_dim must be 128"):
        _lib.check(lib.mmpl_dit_create(C.byref(_dit_cfg(num_heads=4)), C.byref(h)), "mmpl_dit_create")


def test_t5_and_vae_argument_errors():
    lib = _lib.load()
    h = C.c_void_p()
    bad = _lib.MmplT5Config(vocab=100, dim=200, dim_attn=256, dim_ffn=512, num_heads=4, num_layers=1, num_buckets=32, text_len=128, eps=1e-6)
    assert "unsupported dims" in _err(lib.mmpl_t5_create(C.byref(bad), C.byref(h)))
    assert "null argument" in _err(lib.mmpl_t5_create(None, C.byref(h)))
    assert "bad arguments" in _err(lib.mmpl_vae_create(1, 1, C.byref(h)))
    assert "weights not bound" in _err(lib.mmpl_t5_encode(None, None, None, None, None, None, 0, None))
    assert "weights not bound" in _err(lib.mmpl_dit_forward(None, None, None, 1, None, None, None, 0, None, None, 15, None, None, 512, None, None, None, None, None, 0, None))
    assert "shape must be 32 or 16" in _err(lib.mmpl_probe_mfma_tflops(8, 1.0, None))
    assert "null argument" in _err(lib.mmpl_dit_share_check_failures(None, None, None))
    assert lib.mmpl_dit_attn_history_bytes(None, 3) == 0 and lib.mmpl_attn_history_bytes(256 * 3 + 1, 12) == 256 + 12 * 4 * 4 * 128 * 2


def test_kernel_entry_points_reject_bad_shapes():
    lib = _lib.load()
    # K must be a multiple of 64, leading dimensions of 8 elements: rejected before any launch
    assert _err(lib.mmpl_gemm(None, 100, None, 100, None, None, 128, 128, 128, 100, 0, None, 0, None, 0, 1, None))
    kp = (C.c_void_p * 1)(0)
    assert _err(lib.mmpl_attn_fwd(None, 100, None, 128, kp, kp, 128, 128, 1, 64, 64, 1, 0.088, None))      # ldq % 8
    assert _err(lib.mmpl_attn_fwd(None, 128, None, 128, kp, kp, 128, 128, 25, 64, 64, 1, 0.088, None))     # > 24 pages
    assert _err(lib.mmpl_attn_fwd(None, 128, None, 132, kp, kp, 128, 128, 1, 64, 64, 1, 0.088, None))      # ldo % 8 (rows leave 16 B per lane)
    assert _err(lib.mmpl_attn_fwd(None, 128, 8, 128, kp, kp, 128, 128, 1, 64, 64, 1, 0.088, None))         # o not 16-byte aligned
    assert _err(lib.mmpl_attn_fwd_variant(None, 128, None, 128, kp, kp, 128, 128, 1, 64, 64, 1, 0.088, None, 0, 9, 0, None))   # unknown kernel variant
    assert _err(lib.mmpl_attn_fwd_variant(None, 128, None, 128, kp, kp, 128, 128, 1, 64, 64, 1, 0.088, None, 0, 2, 0, None))   # the removed ping-pong kernel
    # mmpl_gemm_scratch: the scratch must be there, large enough and 256-byte aligned (checked before anything touches it)
    nb = lib.mmpl_gemm_scratch_bytes()
    assert nb > 2048
    args = (None, 128, None, 128, None, None, 128, 1024, 256, 128, 0, None, 0, None, 0, 1)
    assert "scratch missing" in _err(lib.mmpl_gemm_scratch(*args, None, nb, None))
    assert "scratch missing" in _err(lib.mmpl_gemm_scratch(*args, C.c_void_p(0x10000), nb - 1, None))
    assert "256-byte aligned" in _err(lib.mmpl_gemm_scratch(*args, C.c_void_p(0x10010), nb, None))
    assert "unknown epilogue" in _err(lib.mmpl_gemm_scratch(*(args[:10] + (99,) + args[11:]), C.c_void_p(0x10000), nb, None))


@pytest.mark.skipif(torch.cuda.is_available(), reason="asserts the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    lib = _lib.load()
    h = C.c_void_p()
    msg = _err(lib.mmpl_dit_create(C.byref(_dit_cfg()), C.byref(h)))
    assert "GPU" in msg or "hip" in msg.lower()
    from mmpl_amd.dit import DitEngine
    from mmpl_amd.synthetic import WAN_CONFIGS
    with pytest.raises(RuntimeError):
        DitEngine(WAN_CONFIGS["tiny"], 16, 24, "cuda:0")
