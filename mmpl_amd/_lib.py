"""ctypes binding of libmmpl_hip.so (include/mmpl_hip.h).  Fails loudly: there is no CPU / PyTorch fallback."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmmpl_hip.so")

# every symbol include/mmpl_hip.h declares (tests/test_abi.py checks this list against the header and the .so)
SYMBOLS = [
    "mmpl_dit_num_weights", "mmpl_dit_weight_name", "mmpl_dit_create", "mmpl_dit_destroy", "mmpl_dit_bind_weights",
    "mmpl_dit_workspace_bytes", "mmpl_dit_context_workspace_bytes", "mmpl_dit_precompute_context", "mmpl_dit_forward", "mmpl_dit_attn_history_bytes", "mmpl_dit_share_check_failures", "mmpl_dit_set_attn_stats", "mmpl_dit_set_image_kv", "mmpl_clip_visual", "mmpl_clip_visual_workspace_bytes",
    "mmpl_attn_fwd", "mmpl_attn_fwd_ws", "mmpl_attn_fwd_variant", "mmpl_attn_fwd_history", "mmpl_attn_history_bytes", "mmpl_attn_workspace_bytes", "mmpl_gemm", "mmpl_gemm_tickets", "mmpl_gemm_scratch", "mmpl_gemm_scratch_bytes", "mmpl_device_xcd_round_robin", "mmpl_probe_mfma_tflops", "mmpl_layernorm", "mmpl_qknorm_rope", "mmpl_cfg_unipc_step", "mmpl_cfg_unipc_step_table",
    "mmpl_vae_num_weights", "mmpl_vae_weight_name", "mmpl_vae_create", "mmpl_vae_destroy", "mmpl_vae_bind_weights",
    "mmpl_vae_workspace_bytes", "mmpl_vae_decode", "mmpl_vae_encode",
    "mmpl_t5_num_weights", "mmpl_t5_create", "mmpl_t5_destroy", "mmpl_t5_bind_weights", "mmpl_t5_workspace_bytes", "mmpl_t5_encode",
    "mmpl_i2v_img_proj_workspace_bytes", "mmpl_i2v_img_proj", "mmpl_i2v_img_kv", "mmpl_i2v_cross_attn_workspace_bytes", "mmpl_i2v_cross_attn",
    "mmpl_profile_enable", "mmpl_profile_read", "mmpl_last_error", "mmpl_version",
]


class MmplDitConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("dim", "ffn_dim", "num_heads", "num_layers", "text_dim", "freq_dim", "in_dim",
                                       "out_dim", "text_len")] + [("eps", C.c_float)] + \
               [(n, C.c_int) for n in ("lat_h", "lat_w", "max_frames")]


class MmplT5Config(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("vocab", "dim", "dim_attn", "dim_ffn", "num_heads", "num_layers", "num_buckets", "text_len")] + \
               [("eps", C.c_float)]


class MmplUniPCStep(C.Structure):
    _fields_ = [("guidance", C.c_float), ("sigma_cur", C.c_float), ("use_corrector", C.c_int), ("corr_order", C.c_int),
                ("c_c1", C.c_float), ("c_c2", C.c_float), ("c_c3", C.c_float), ("c_inv_rk", C.c_float),
                ("c_rho0", C.c_float), ("c_rho_last", C.c_float), ("pred_order", C.c_int), ("p_c1", C.c_float),
                ("p_c2", C.c_float), ("p_c3", C.c_float), ("p_inv_rk", C.c_float)]


_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built. Run `python -m mmpl_amd.build` "
            "(or __graft_entry__.build()). mmpl_amd has no CPU/PyTorch fallback by design.")
    # torch first: it ships its own libamdhip64, and device memory / streams are shared with it.  If this library were
    # dlopen'ed before torch, the system HIP runtime would be bound instead and every hipMalloc here would fail once torch
    # initialises the device through the other copy (seen as "hipMalloc ... failed" in build() -> smoke() in one process).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    lib.mmpl_last_error.restype = C.c_char_p
    lib.mmpl_version.restype = C.c_char_p
    lib.mmpl_dit_num_weights.argtypes = [C.POINTER(MmplDitConfig)]
    lib.mmpl_dit_weight_name.argtypes = [ci, ci]
    lib.mmpl_dit_weight_name.restype = C.c_char_p
    lib.mmpl_dit_create.argtypes = [C.POINTER(MmplDitConfig), C.POINTER(vp)]
    lib.mmpl_dit_destroy.argtypes = [vp]
    lib.mmpl_dit_destroy.restype = None
    lib.mmpl_dit_bind_weights.argtypes = [vp, C.POINTER(vp), ci]
    lib.mmpl_dit_workspace_bytes.argtypes = [vp, ci]
    lib.mmpl_dit_workspace_bytes.restype = sz
    lib.mmpl_dit_context_workspace_bytes.argtypes = [vp]
    lib.mmpl_dit_context_workspace_bytes.restype = sz
    lib.mmpl_dit_precompute_context.argtypes = [vp, vp, vp, vp, vp, sz, C.POINTER(ci), vp]
    lib.mmpl_dit_forward.argtypes = [vp, vp, vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci), ci, vp, vp, ci, vp, vp,
                                     ci, vp, vp, vp, vp, vp, sz, vp]
    lib.mmpl_dit_attn_history_bytes.argtypes = [vp, ci]
    lib.mmpl_dit_attn_history_bytes.restype = sz
    lib.mmpl_dit_share_check_failures.argtypes = [vp, C.POINTER(C.c_longlong), vp]
    lib.mmpl_attn_history_bytes.argtypes = [ci, ci]
    lib.mmpl_attn_history_bytes.restype = sz
    lib.mmpl_attn_fwd_history.argtypes = [vp, ci, vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ci, ci, ci, ci, ci, cf, vp, sz, vp, vp, vp]
    lib.mmpl_attn_fwd.argtypes = [vp, ci, vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ci, ci, ci, ci, ci, cf, vp]
    lib.mmpl_attn_fwd_ws.argtypes = [vp, ci, vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ci, ci, ci, ci, ci, cf, vp, sz, vp]
    lib.mmpl_attn_fwd_variant.argtypes = [vp, ci, vp, ci, C.POINTER(vp), C.POINTER(vp), ci, ci, ci, ci, ci, ci, cf, vp, sz, ci, ci, vp]
    lib.mmpl_cfg_unipc_step_table.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, ci, ci, vp]
    lib.mmpl_attn_workspace_bytes.argtypes = []
    lib.mmpl_attn_workspace_bytes.restype = sz
    lib.mmpl_gemm.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp]
    lib.mmpl_gemm_tickets.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, vp]
    lib.mmpl_gemm_scratch.argtypes = [vp, ci, vp, ci, vp, vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, ci, vp, sz, vp]
    lib.mmpl_gemm_scratch_bytes.argtypes = []
    lib.mmpl_device_xcd_round_robin.argtypes = []
    lib.mmpl_probe_mfma_tflops.argtypes = [ci, C.c_double, C.POINTER(C.c_double)]
    lib.mmpl_gemm_scratch_bytes.restype = sz
    lib.mmpl_layernorm.argtypes = [vp, ci, vp, ci, ci, ci, cf, vp, vp, ci, ci, vp, vp, vp]
    lib.mmpl_qknorm_rope.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, ci, C.POINTER(ci), C.POINTER(vp), C.POINTER(vp), vp]
    lib.mmpl_cfg_unipc_step.argtypes = [vp, vp, vp, vp, vp, vp, sz, C.POINTER(MmplUniPCStep), vp]
    lib.mmpl_profile_enable.argtypes = [ci]
    lib.mmpl_profile_read.argtypes = [ci, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    lib.mmpl_t5_num_weights.argtypes = [C.POINTER(MmplT5Config)]
    lib.mmpl_t5_create.argtypes = [C.POINTER(MmplT5Config), C.POINTER(vp)]
    lib.mmpl_t5_destroy.argtypes = [vp]
    lib.mmpl_t5_destroy.restype = None
    lib.mmpl_t5_bind_weights.argtypes = [vp, C.POINTER(vp), ci]
    lib.mmpl_t5_workspace_bytes.argtypes = [vp]
    lib.mmpl_t5_workspace_bytes.restype = sz
    lib.mmpl_t5_encode.argtypes = [vp, vp, vp, vp, vp, vp, sz, vp]
    cf = C.c_float
    lib.mmpl_i2v_img_proj_workspace_bytes.argtypes = [ci, ci, ci]
    lib.mmpl_i2v_img_proj_workspace_bytes.restype = sz
    lib.mmpl_dit_set_image_kv.argtypes = [vp, vp, vp, ci]
    lib.mmpl_dit_set_attn_stats.argtypes = [vp, vp]
    lib.mmpl_clip_visual_workspace_bytes.argtypes = [ci, ci, ci, ci]
    lib.mmpl_clip_visual_workspace_bytes.restype = sz
    lib.mmpl_clip_visual.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, C.POINTER(vp), C.POINTER(vp), cf, vp, vp, sz, vp]
    lib.mmpl_i2v_img_proj.argtypes = [vp, ci, ci, ci, C.POINTER(vp), vp, vp, sz, vp]
    lib.mmpl_i2v_img_kv.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, vp]
    lib.mmpl_i2v_cross_attn_workspace_bytes.argtypes = [ci, ci]
    lib.mmpl_i2v_cross_attn_workspace_bytes.restype = sz
    lib.mmpl_i2v_cross_attn.argtypes = [vp, ci, ci, vp, vp, vp, cf, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, sz, vp]
    if hasattr(lib, "mmpl_vae_create"):
        _bind_vae(lib)
    _lib = lib
    return lib


def _bind_vae(lib):
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    fp = C.POINTER(C.c_float)
    lib.mmpl_vae_weight_name.argtypes = [ci]
    lib.mmpl_vae_weight_name.restype = C.c_char_p
    lib.mmpl_vae_create.argtypes = [ci, ci, C.POINTER(vp)]
    lib.mmpl_vae_destroy.argtypes = [vp]
    lib.mmpl_vae_destroy.restype = None
    lib.mmpl_vae_bind_weights.argtypes = [vp, C.POINTER(vp), ci]
    lib.mmpl_vae_workspace_bytes.argtypes = [vp, ci]
    lib.mmpl_vae_workspace_bytes.restype = sz
    lib.mmpl_vae_decode.argtypes = [vp, vp, ci, fp, fp, vp, vp, sz, vp]
    lib.mmpl_vae_encode.argtypes = [vp, vp, ci, fp, fp, vp, vp, sz, vp]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libmmpl_hip {what}: {load().mmpl_last_error().decode()}")


def ptr(t) -> C.c_void_p:
    """Borrowed device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def stream_ptr() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
