"""ctypes binding of libaether_hip.so (include/aether_hip.h).

The product path has NO fallback: if the shared object is missing or an entry point reports an error, a
RuntimeError/ValueError is raised.  Nothing under oracle/ is imported from here.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libaether_hip.so"
_lib = None

AETHER_EPI_BIAS = 0
AETHER_EPI_BIAS_GELU = 1
AETHER_EPI_BIAS_GATE_RES = 2
AETHER_GEMM_WIDE_STORE = 1
AETHER_ATTN_EXACT_MAX = 32    # attention: conservative path only (true-maximum shift from tile 0, a-posteriori check per tile)
AETHER_VAE_TWO_LANES = 256    # VAE plan: tile batches of two on two streams (see include/aether_hip.h)
AETHER_CONV_TAP_REUSE = 128   # conv: K order is (dt, dh, channel block, dw) -> the tap-reuse kernel may be used
ATTN_Q_SCALE = 0.125 * 1.4426950408889634   # softmax scale x log2(e): the attention kernel works in the log2 domain
PROF_CLASSES = ["other", "layernorm", "gemm_qkv", "qk_norm_rope", "attention", "gemm_out", "gemm_ff1", "gemm_ff2"]

_vp, _i, _f, _fp, _sz = C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_size_t


class AetherDitConfig(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int), ("num_heads", C.c_int), ("head_dim", C.c_int), ("in_channels", C.c_int),
        ("out_channels", C.c_int), ("patch_size", C.c_int), ("text_dim", C.c_int), ("time_embed_dim", C.c_int),
        ("ff_mult", C.c_int), ("max_text_len", C.c_int), ("norm_eps", C.c_float), ("qk_norm_eps", C.c_float),
        ("use_pos_embedding", C.c_int), ("flags", C.c_int),
    ]


class AetherVaeConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int), ("layers_per_block", C.c_int),
        ("num_levels", C.c_int), ("norm_num_groups", C.c_int), ("temporal_compression_ratio", C.c_int), ("sample_height", C.c_int),
        ("sample_width", C.c_int), ("norm_eps", C.c_float), ("tap_reuse_max_waste", C.c_float), ("flags", C.c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/aether_hip.h
SIGNATURES = {
    "aether_last_error": (C.c_char_p, []),
    "aether_version": (_i, []),
    "aether_check_device": (_i, []),
    "aether_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _fp, _i, _vp, _i, _fp, _fp, _i, _i, _i, _fp, _sz, _i, _vp]),
    "aether_layernorm_modulate": (_i, [_vp, _i, _vp, _i, _i, _i, _f, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _vp]),
    "aether_gemv_rows": (_i, [_fp, _i, _i, _vp, _fp, _fp, _i, _i, _i, _vp]),
    "aether_timestep_sinusoid": (_i, [_fp, _i, _i, _fp, _vp]),
    "aether_patchify": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aether_unpatchify": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aether_qk_norm_rope": (_i, [_vp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _f, _fp, _fp, _f, _vp, _vp, _vp, _i, _vp]),
    "aether_flash_attn_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "aether_dpm_step": (_i, [_vp, _i, _f, _vp, _fp, _vp, _f, _f, _f, _f, _f, _f, _f, _fp, _fp, _vp, C.c_long, _vp]),
    "aether_conv_gemm_bf16": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _i, _fp, _vp, _i, _fp, _sz, _i, _vp]),
    "aether_im2col_first": (_i, [_vp, C.c_long, C.c_long, C.c_long, C.c_long, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "aether_groupnorm_stats": (_i, [_vp, _i, _i, _i, _i, _f, _fp, _fp, _fp, _i, _fp, _fp, _vp]),
    "aether_spatial_cond": (_i, [_vp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _fp, _vp]),
    "aether_groupnorm_apply": (_i, [_vp, _i, _i, _i, _i, _i, _fp, _i, _vp, _i, _i, _i, _i, _i, _i, _fp, _i, _i, _i, C.POINTER(C.c_int), _vp]),
    "aether_groupnorm_apply_causal": (_i, [_vp, _i, _i, _i, _i, _i, _fp, _i, _vp, _i, _i, _i, _i, _fp, _i, _i, _i, C.POINTER(C.c_int), _vp, _vp, _vp]),
    "aether_causal_front": (_i, [_vp, _i, _i, C.c_long, _vp, _vp, _vp]),
    "aether_resample_pad": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aether_preprocess_frames": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "aether_merge_scale_fit": (_i, [_fp, _vp, C.c_long, _vp, _i, _vp, _vp]),
    "aether_merge_window": (_i, [_fp, _fp, _vp, _vp, _i, _i, C.c_long, C.POINTER(C.c_double), _vp, _vp]),
    "aether_backproject": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "aether_vae_create": (_vp, [C.POINTER(AetherVaeConfig)]),
    "aether_vae_destroy": (None, [_vp]),
    "aether_vae_set_conv": (_i, [_vp, C.c_char_p, _vp, _fp, _i, _i, _i, _i, _i, _i, _i, _i]),
    "aether_vae_set_norm": (_i, [_vp, C.c_char_p, _fp, _fp, _fp, _fp, _fp, _fp]),
    "aether_vae_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "aether_vae_output_shape": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "aether_vae_encode": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "aether_vae_decode": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "aether_dit_create": (_vp, [C.POINTER(AetherDitConfig)]),
    "aether_dit_destroy": (None, [_vp]),
    "aether_dit_set_weight": (_i, [_vp, C.c_char_p, _vp]),
    "aether_dit_set_pos_embedding": (_i, [_vp, _vp, _i]),
    "aether_dit_set_flags": (_i, [_vp, _i]),
    "aether_dit_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "aether_dit_set_profile": (_i, [_vp, _i]),
    "aether_dit_get_profile": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "aether_dit_forward": (_i, [_vp, _vp, _vp, _fp, _fp, _fp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
}


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libaether_hip.so (built by aether_amd.build.build_native / __graft_entry__.build). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built (run `python -m aether_amd.build`). "
            "aether_amd has no CPU fallback."
        )
    # PyTorch ships its own libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's).  Whichever copy is mapped first serves
    # the whole process, and the kernels must run in the runtime instance that owns torch's allocations and streams: import
    # torch BEFORE dlopen-ing the library (loading /opt/rocm's runtime first and torch afterwards leaves hipGetDevice failing).
    import torch  # noqa: F401

    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here means the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc == 0:
        return
    msg = load().aether_last_error().decode("utf-8", "replace")
    if rc in (-1, -2, -3):
        raise ValueError(f"{what}: {msg} (code {rc})")
    raise RuntimeError(f"{what}: {msg} (code {rc})")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
