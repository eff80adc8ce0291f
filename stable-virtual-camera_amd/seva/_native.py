"""ctypes binding of libseva_hip.so (C-ABI declared in include/seva_hip.h).

There is NO fallback: if the shared library is missing or a symbol cannot be resolved the
import of the HIP-backed operators fails loudly (`SevaNativeError`).  Nothing in this package
computes the hot path with PyTorch ops.
"""

from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEVA_HIP_LIB: A/B benchmarking of two builds of the same library (tools/); default = the in-tree build
LIB_PATH = os.environ.get("SEVA_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libseva_hip.so")
ABI_VERSION = 8
PROF_CLASSES = 5
PROF_NAMES = ("gemm", "conv", "attention", "norm", "elementwise")


class SevaNativeError(RuntimeError):
    pass


class GemmDesc(C.Structure):
    _fields_ = [
        ("a", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("row_add", c_void_p),
        ("residual", c_void_p), ("out_f32", c_void_p), ("out_f16", c_void_p),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("lda", c_int64), ("ldr", c_int64), ("ldo32", c_int64), ("ldo16", c_int64),
        ("rows_per_group", c_int64), ("ld_row_add", c_int64),
        ("mode", c_int32), ("epilogue", c_int32),
        ("n", c_int32), ("ih", c_int32), ("iw", c_int32), ("cin", c_int32),
        ("oh", c_int32), ("ow", c_int32), ("stride", c_int32), ("upsample", c_int32),
        ("col_scale", c_float), ("col_scale_n", c_int32), ("pad_br_only", c_int32),
        ("w_exp", c_void_p), ("out_f8", c_void_p), ("ldo8", c_int64),
        ("ch_stats", c_void_p), ("splitk_ws", c_void_p), ("splitk_ws_bytes", c_int64),
        ("a2", c_void_p), ("lda2", c_int64), ("K2", c_int64),
        ("alg_K", c_int64),
    ]


class FfDesc(C.Structure):
    _fields_ = [
        ("a", c_void_p), ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p),
        ("residual", c_void_p), ("out_f32", c_void_p), ("out_f16", c_void_p),
        ("M", c_int64), ("lda", c_int64), ("ldr", c_int64), ("ldo32", c_int64), ("ldo16", c_int64),
        ("C", c_int32),
        ("ln_x", c_void_p), ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ldx", c_int64), ("ln_eps", c_float),
    ]


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p),
        ("q_sb0", c_int64), ("q_sb1", c_int64), ("q_sl", c_int64),
        ("k_sb0", c_int64), ("k_sb1", c_int64), ("k_sl", c_int64),
        ("o_sb0", c_int64), ("o_sb1", c_int64), ("o_sl", c_int64),
        ("nb0", c_int32), ("nb1", c_int32), ("heads", c_int32),
        ("lq", c_int32), ("lk", c_int32), ("scale", c_float), ("q_prescaled", c_int32),
        ("split_ws", c_void_p), ("split_ws_bytes", c_int64),
    ]


class GroupNormDesc(C.Structure):
    _fields_ = [
        ("x1", c_void_p), ("x2", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
        ("dense", c_void_p), ("dense_w", c_void_p), ("dense_b", c_void_p),
        ("out_f16", c_void_p), ("workspace", c_void_p),
        ("n", c_int32), ("hw", c_int32), ("c1", c_int32), ("c2", c_int32),
        ("groups", c_int32), ("dense_c", c_int32), ("silu", c_int32), ("eps", c_float),
        ("raw_f16", c_void_p), ("out_f8", c_void_p), ("ld_out_f8", c_int64),
        ("stats1", c_void_p), ("stats2", c_void_p),
        ("split_out_f16", c_int32), ("split_raw_f16", c_int32),
    ]


# name -> (restype, argtypes); must list every symbol declared in include/seva_hip.h
SYMBOLS = {
    "seva_last_error": (c_char_p, []),
    "seva_abi_version": (c_int, []),
    "seva_target_arch": (c_char_p, []),
    "seva_gemm_f16": (c_int, [POINTER(GemmDesc), c_void_p]),
    "seva_gemm_fp8": (c_int, [POINTER(GemmDesc), c_void_p]),
    "seva_ff_fused_f16": (c_int, [POINTER(FfDesc), c_void_p]),
    "seva_attention_f16": (c_int, [POINTER(AttnDesc), c_void_p]),
    "seva_groupnorm_f16": (c_int, [POINTER(GroupNormDesc), c_void_p]),
    "seva_layernorm_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "seva_layernorm_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_int64, c_void_p]),
    "seva_layernorm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p]),
    "seva_clip_preprocess_f16": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p, c_int32, c_void_p]),
    "seva_attention_small_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                         c_int64, c_int64, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    "seva_softmax_rows_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int32, c_int32, c_float, c_void_p]),
    "seva_nchw_to_nhwc_f16": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_nchw_to_nhwc_f16_split": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_cast_concat_f16": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    "seva_bilinear_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "seva_timestep_embedding_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    "seva_silu_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "seva_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "seva_replace_blend_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_denoiser_combine_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_add_noise_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_cfg_euler_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_cfg_combine_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_euler_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_to_d_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_scale_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p]),
    "seva_plucker_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_cond_concat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "seva_set_knob": (c_int, [c_char_p, c_int32]),
    "seva_get_knob": (c_int, [c_char_p, POINTER(c_int32)]),
    "seva_graph_begin": (c_int, [c_void_p]),
    "seva_graph_end": (c_int, [c_void_p, POINTER(c_void_p)]),
    "seva_graph_launch": (c_int, [c_void_p, c_void_p]),
    "seva_graph_destroy": (c_int, [c_void_p]),
    "seva_prof_enable": (c_int, [c_int]),
    "seva_prof_collect": (c_int, [POINTER(c_double), POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
}

_lib = None

# > 0: `SevaEngine.__call__` launches the network eagerly instead of replaying its own network-only hipGraph
# (set around the warm-up step of the whole-step graph, seva/_stepgraph.py)
_EAGER_DEPTH = 0


class eager_network:
    def __enter__(self):
        global _EAGER_DEPTH
        _EAGER_DEPTH += 1

    def __exit__(self, *exc):
        global _EAGER_DEPTH
        _EAGER_DEPTH -= 1
        return False


def network_eager_forced() -> bool:
    return _EAGER_DEPTH > 0


def load() -> C.CDLL:
    """Load libseva_hip.so and bind every symbol; raises SevaNativeError on any problem."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SevaNativeError(
            f"{LIB_PATH} not found: build it with `make -C stable-virtual-camera_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise SevaNativeError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SevaNativeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.seva_abi_version() != ABI_VERSION:
        raise SevaNativeError(
            f"ABI mismatch: library {lib.seva_abi_version()} vs binding {ABI_VERSION}"
        )
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().seva_last_error().decode("utf-8", "replace")
        raise SevaNativeError(f"{what or 'seva native call'} failed (rc={rc}): {msg}")


def stream_ptr(device=None) -> int:
    """Raw hipStream_t of PyTorch's current stream (kernels launch on it)."""
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise SevaNativeError(
                "seva HIP operators need tensors on an AMD GPU (cuda device); "
                "there is no CPU fallback path."
            )
