"""ctypes binding of libucfvit_hip.so (C ABI declared in include/ucfvit_hip.h).

The library is the product: there is no CPU or PyTorch fallback behind these calls.  If the shared object is
missing or a call fails, a RuntimeError is raised with the library's own message.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

F32, BF16 = 0, 1
LAYOUT_KC, LAYOUT_KS = 0, 1
ACT_NONE, ACT_GELU, ACT_GELU_GRAD, ACT_GELU_SAVE_DERIV, ACT_MUL_AUX = 0, 1, 2, 3, 4
GEMM_SCHED_BYTES = 1024
ABI_VERSION = 12

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # .../ucf-vit_amd
# UCFVIT_HIP_LIB: an alternative build of the same library (A/B measurements of kernel variants); never a non-HIP fallback
LIB_PATH = os.environ.get("UCFVIT_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libucfvit_hip.so")


class GemmDesc(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("residual", c_void_p), ("aux_in", c_void_p), ("aux_out", c_void_p),
        ("M", c_int64), ("N", c_int64), ("K", c_int64),
        ("lda", c_int64), ("ldb", c_int64), ("ldc", c_int64), ("ldr", c_int64), ("ldaux", c_int64),
        ("a_layout", c_int32), ("b_layout", c_int32),
        ("dtype", c_int32), ("out_dtype", c_int32),
        ("act", c_int32), ("accumulate", c_int32),
        ("alpha", c_float),
        ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("c_colsum_partial", c_void_p),
        ("sched_state", c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol of include/ucfvit_hip.h (tests/test_abi.py checks this)
_P, _I64, _I, _F = c_void_p, c_int64, c_int, c_float
SIGNATURES = {
    "ucfvit_abi_version": (c_int, []),
    "ucfvit_mfma_probe": (_I64, [_P, _I, _P]),
    "ucfvit_occupy": (c_int, [_I, _I, _P, _P]),
    "ucfvit_last_error": (c_char_p, []),
    "ucfvit_gemm_workspace": (c_int64, [POINTER(GemmDesc)]),
    "ucfvit_gemm": (c_int, [POINTER(GemmDesc), _P]),
    "ucfvit_gemm_colsum_rows": (c_int64, [POINTER(GemmDesc)]),
    "ucfvit_reduce_rows": (c_int, [_P, _P, _I64, _I64, _I, _P]),
    "ucfvit_gemm_grouped": (c_int, [POINTER(GemmDesc), _I64, _P]),
    "ucfvit_colsum_workspace": (c_int64, [_I64, _I64]),
    "ucfvit_colsum": (c_int, [_P, _P, _I64, _I64, _I64, _I, _P, _I, _P]),
    "ucfvit_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _I, _P]),
    "ucfvit_layernorm_bwd_workspace": (c_int64, [_I64, _I64]),
    "ucfvit_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _P, _I, _P, _I, _P]),
    "ucfvit_attention_fwd": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_attention_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_attention_bwd_colsum_supported": (c_int, [_I64, _I64, _I64, _I64, _I]),
    "ucfvit_attention_bwd_colsum": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_attention_cross_fwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_attention_cross_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _I, _I, _P]),
    "ucfvit_attention_merge": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _I, _P]),
    "ucfvit_im2col": (c_int, [_P, _P, _I64, _I64, POINTER(c_int64), _I, _I64, _I, _P]),
    "ucfvit_tokens_fwd": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _P]),
    "ucfvit_tokens_bwd": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _P]),
    "ucfvit_seq_patches": (c_int, [_P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "ucfvit_varagg_fwd": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_varagg_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_quadtree_workspace": (_I64, [_I64, _I64, _I64]),
    "ucfvit_octree_workspace": (_I64, [_I64, _I64]),
    "ucfvit_octree_build": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _P, _P]),
    "ucfvit_octree_serialize": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _P]),
    "ucfvit_quadtree_build": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P, _P]),
    "ucfvit_quadtree_serialize": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P]),
    "ucfvit_adaptive_pos_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _P]),
    "ucfvit_adaptive_pos_bwd_workspace": (_I64, [_I64, _I64, _I64, _I, _I, _I]),
    "ucfvit_adaptive_pos_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I, _I, _I, _P, _I, _P]),
    "ucfvit_instnorm_workspace": (_I64, [_I64, _I64]),
    "ucfvit_instnorm_fwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _F, _F, _P, _I, _P]),
    "ucfvit_instnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _F, _P, _I, _P]),
    "ucfvit_dice_ce_workspace": (_I64, [_I64, _I64]),
    "ucfvit_dice_ce": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _F, _F, _F, _P, _I, _P]),
    "ucfvit_dice_ce_strided": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _F, _F, _F, _P, _I, _P]),
    "ucfvit_dice_ce_stats_floats": (c_int, []),
    "ucfvit_dice_ce_stats": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _I, _P]),
    "ucfvit_dice_ce_from_stats": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _F, _F, _F, _I, _P]),
    "ucfvit_instnorm_cl_bwd_sums": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P, _P]),
    "ucfvit_instnorm_cl_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P]),
    "ucfvit_instnorm_cl_workspace": (_I64, [_I64, _I64, _I64]),
    "ucfvit_instnorm_cl_fwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _F, _P, _P]),
    "ucfvit_instnorm_cl_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _I, _P, _P]),
    "ucfvit_instnorm_cl_stats": (c_int, [_P, _P, _P, _I64, _I64, _I64, _F, _P, _P]),
    "ucfvit_instnorm_cl_apply": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _P]),
    "ucfvit_instnorm_cl_apply2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _F, _P]),
    "ucfvit_instnorm_cl_bwd2_workspace": (_I64, [_I64, _I64, _I64]),
    "ucfvit_instnorm_cl_bwd2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _F, _P, _P]),
    "ucfvit_conv3d_fwd": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _I64, _I64, _I, _I, _P, _P]),
    "ucfvit_conv3d_fwd_stats_rows": (_I64, [_I64, _I64, _I64, _I64, _I64, _I64, _I, _I]),
    "ucfvit_instnorm_cl_stats_fold": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _F, _P, _P]),
    "ucfvit_conv3d_wgrad_size": (_I64, [_I64, _I64, _I]),
    "ucfvit_conv3d_wgrad_workspace": (_I64, [_I64, _I64, _I64, _I64, _I64, _I64, _I]),
    "ucfvit_conv3d_wgrad": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "ucfvit_depth_to_space2": (c_int, [_P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I, _P, _I64, _P]),
    "ucfvit_pad_channels8": (c_int, [_P, _P, _I64, _I64, _I64, _P]),
    "ucfvit_pad_rows8": (c_int, [_P, c_int, _P, _I64, _I64, _I64, _P]),
    "ucfvit_cross_entropy": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _F, _I, _P]),
    "ucfvit_mae_mask": (c_int, [_P, _P, _P, _P, _I64, _I64, _I64, _P]),
    "ucfvit_gather_rows": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "ucfvit_scatter_rows": (c_int, [_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I, _P]),
    "ucfvit_unshuffle_fwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P]),
    "ucfvit_unshuffle_bwd_workspace": (c_int64, [_I64, _I64]),
    "ucfvit_unshuffle_bwd": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I, _P, _I, _P]),
    "ucfvit_patch_mse": (c_int, [_P, _P, _P, _P, _P, _I64, _I64, POINTER(c_int64), _I, _I64, _F, _P, _I, _P]),
    "ucfvit_adamw": (c_int, [_P, _P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _F, _I, _P]),
    "ucfvit_cast": (c_int, [_P, _P, _I64, _I, _I, _F, _P]),
    "ucfvit_transpose_batched": (c_int, [_P, _P, _P, _I64, _I64, _P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Fails loudly: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C {_PKG_ROOT}` (or __graft_entry__.build()). "
            "The UCF_VIT operators have no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    v = lib.ucfvit_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"libucfvit_hip.so ABI version {v} != binding version {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ucfvit_last_error()
        raise HipLibraryError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
