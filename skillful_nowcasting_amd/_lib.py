"""ctypes binding of libdgmr_hip.so (C ABI declared in include/dgmr_hip.h).

The product path has no fallback: if the shared library is missing or a kernel call fails, a
``RuntimeError`` is raised.  Nothing here (or anywhere in this package) imports ``oracle/``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGMR_LIB=<path>: load another build of the library (A/B runs of a compiler flag or an experimental kernel; the tests use the default)
LIB_PATH = os.environ.get("DGMR_LIB") or os.path.join(_HERE, "lib", "libdgmr_hip.so")
ABI_VERSION = 11

P = c_void_p  # every device pointer and the stream travel as void*
# deterministic mode (fixed-order cross-workgroup sums: bit-identical runs) is the default; DGMR_DETERMINISTIC=0 switches it off (A/B)
DETERMINISTIC_DEFAULT = os.environ.get("DGMR_DETERMINISTIC", "1") != "0"


class ConvArgs(Structure):
    """Mirror of ``dgmr_conv_args`` (include/dgmr_hip.h)."""

    _fields_ = [
        ("x", P), ("w", P), ("bias", P), ("scale", P), ("pre_a", P), ("pre_b", P), ("addend", P), ("residual", P),
        ("mask_src", P), ("mask_a", P), ("mask_b", P), ("y", P),
        ("N", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
        ("KD", c_int32), ("KH", c_int32), ("KW", c_int32), ("upsample", c_int32), ("pre_relu", c_int32),
        ("scale_group", c_int32), ("pre_group", c_int32), ("mask_group", c_int32), ("act_relu", c_int32),
        ("w_cin", c_int32), ("w_coff", c_int32), ("epi_mode", c_int32), ("ksplit", c_int32),
        ("gru_h", P), ("gru_pu", P), ("pre_out", P), ("splitk_ws", P), ("splitk_ws_bytes", c_int64), ("w_split", P),
        ("residual_up", c_int32), ("reserved0", c_int32), ("stats_out", P), ("w_phase", P), ("pool2", c_int32), ("reserved1", c_int32),
        ("scale2", P), ("bias2", P), ("addend2", P), ("y2", P), ("gru_split", c_int32), ("reserved2", c_int32),
    ]


class WgradArgs(Structure):
    """Mirror of ``dgmr_wgrad_args``."""

    _fields_ = [
        ("x", P), ("dy", P), ("pre_a", P), ("pre_b", P), ("partial", P),
        ("N", c_int32), ("D", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
        ("KD", c_int32), ("KH", c_int32), ("KW", c_int32), ("upsample", c_int32), ("pre_relu", c_int32),
        ("pre_group", c_int32), ("nsplit", c_int32), ("groups", c_int32), ("bias_grad", P),
        ("bias_partial", P), ("bias_rows", c_int32), ("bias_stride", c_int32),  # ABI 11: deterministic bias gradient
    ]


class SNDesc(Structure):
    """Mirror of ``dgmr_sn_desc``."""

    _fields_ = [
        ("w", P), ("gram", P), ("u", P), ("v", P),
        ("inv_sigma_off", c_int64), ("u_hist_off", c_int64), ("v_hist_off", c_int64), ("tmp_off", c_int64),
        ("Cout", c_int32), ("Cin", c_int32), ("taps", c_int32), ("T", c_int32), ("eps", c_float),
        ("row_block0", c_int32), ("col_block0", c_int32), ("iter_block0", c_int32), ("perm", P),
    ]


class AdamDesc(Structure):
    """Mirror of ``dgmr_adam_desc``."""

    _fields_ = [("p", P), ("g", P), ("m", P), ("v", P), ("n", c_int64), ("block0", c_int32), ("step_size", c_float), ("bc2_sqrt", c_float),
                ("reserved", c_int32)]


def _adam_desc_dtype():
    import numpy as np

    return np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("block0", "<i4"), ("step_size", "<f4"),
                     ("bc2_sqrt", "<f4"), ("reserved", "<i4")])


ADAM_DESC_DTYPE = _adam_desc_dtype()  # the same 56 bytes as a numpy record (descriptor tables are filled on the host with numpy)
assert ADAM_DESC_DTYPE.itemsize == 56

i, f, L = c_int, c_float, c_int64
# name -> argtypes (every function returns int except the two noted below); must match include/dgmr_hip.h
SIGNATURES = {
    "dgmr_conv_fwd": [POINTER(ConvArgs), P],
    "dgmr_conv_stats_rows": [POINTER(ConvArgs)],
    "dgmr_bn_partial_reduce": [P, P, i, L, i, P],
    "dgmr_bn_bwd_center": [P, P, P, i, i, P],
    "dgmr_conv_flip_weights": [P, P, i, i, i, i, i, i, i, P],
    "dgmr_conv_wgrad": [POINTER(WgradArgs), P],
    "dgmr_conv_wgrad_nsplit": [i, i, i, i],
    "dgmr_conv_wgrad_plan": [POINTER(WgradArgs)],
    "dgmr_wgrad_reduce": [P, i, i, L, P, P, P, P, P],
    "dgmr_wgrad_reduce_slice": [P, i, i, i, i, i, i, i, P, P, P, P, P],
    "dgmr_sn_wgrad_finalize": [P, P, P, P, P, P, i, i, i, i, i, P],
    "dgmr_spectral_sigma": [P, P, P, P, P, P, P, P, i, i, i, f, i, P],
    "dgmr_spectral_sigma_seq": [P, P, P, P, P, P, P, P, P, i, i, i, f, i, P],
    "dgmr_spectral_sigma_seq_multi": [P, i, i, i, i, i, i, P, P],
    "dgmr_bn_stats": [P, P, i, L, i, P],
    "dgmr_bn_finalize": [P, P, P, P, P, P, P, P, P, P, i, L, i, f, f, P, P],
    "dgmr_bn_bwd_reduce": [P, P, P, P, P, i, L, i, P],
    "dgmr_bn_bwd_apply": [P, P, P, P, P, P, P, P, P, P, i, L, i, i, P],
    "dgmr_colsum": [P, P, P, L, i, i, P],
    "dgmr_affine": [P, P, P, P, i, L, i, i, P],
    "dgmr_pool_fwd": [P, P, P, i, i, i, i, i, i, f, P, P, P, i, P],
    "dgmr_pool_bwd": [P, P, i, i, i, i, i, i, f, P],
    "dgmr_pool_depth2": [P, P, P, i, i, L, P],
    "dgmr_frames_s2d": [P, P, P, i, i, i, i, i, i, i, i, i, P],
    "dgmr_frames_s2d_bwd": [P, P, P, i, i, i, i, i, i, i, i, i, P],
    "dgmr_d2s_frames": [P, P, i, i, i, i, i, i, P],
    "dgmr_d2s_frames_bwd": [P, P, i, i, i, i, i, i, P],
    "dgmr_permute_nt": [P, P, i, i, L, P],
    "dgmr_copy_channels": [P, P, L, i, i, i, i, i, i, i, i, P],
    "dgmr_gru_gate_fwd": [P, P, P, L, P],
    "dgmr_gru_gate_bwd": [P, P, P, P, P, L, P],
    "dgmr_gru_blend_fwd": [P, P, P, P, L, P],
    "dgmr_gru_blend_bwd": [P, P, P, P, P, P, P, L, P],
    "dgmr_axpby": [P, P, P, f, f, L, P],
    "dgmr_repeat_rows": [P, P, L, i, P],
    "dgmr_group_rowsum": [P, P, P, i, i, L, i, i, P],
    "dgmr_repeat_interleave": [P, P, L, L, i, P],
    "dgmr_scale_by_dev": [P, P, f, P, L, P],
    "dgmr_relu_bwd": [P, P, P, L, P],
    "dgmr_fill": [P, f, L, P],
    "dgmr_attention_fwd": [P, P, P, P, P, i, i, i, P],
    "dgmr_attention_bwd": [P, P, P, P, P, P, P, P, P, i, i, i, P],
    "dgmr_relu_sum_hw_fwd": [P, P, i, i, i, P],
    "dgmr_relu_sum_hw_bwd": [P, P, P, i, i, i, P],
    "dgmr_linear1_fwd": [P, P, P, P, P, i, i, i, P],
    "dgmr_linear1_bwd": [P, P, P, P, P, P, P, i, i, i, P],
    "dgmr_hinge_disc": [P, P, P, P, P, i, i, P],
    "dgmr_grid_cell_loss": [P, i, L, P, P, f, P, P, f, P, L, P],
    "dgmr_adam": [P, P, P, P, L, c_double, c_double, c_double, c_double, i, P],
    "dgmr_adam_chunk": [],
    "dgmr_adam_multi": [P, i, i, c_double, c_double, c_double, P],
    "dgmr_upsample_phase_weights": [P, P, i, i, P],
    "dgmr_pool2_phase_weights": [P, P, i, i, P],
    "dgmr_upsample_wgrad_sums": [P, P, i, i, i, i, P],
    "dgmr_head_blocks": [L, L, i],
    "dgmr_head_fwd": [P, P, P, P, P, P, P, L, L, i, P],
    "dgmr_head_bwd_sums": [P, P, P, P, P, P, P, P, P, L, L, i, P],
    "dgmr_head_bwd_apply": [P, P, P, P, P, P, P, P, P, P, P, P, P, L, L, i, i, P],
    "dgmr_conv_pool2_supported": [POINTER(ConvArgs)],
    "dgmr_split_weights": [P, P, L, i, i, i, i, L, P],
    "dgmr_conv_gates2_supported": [POINTER(ConvArgs)],
    "dgmr_set_precision": [i],
    "dgmr_set_deterministic": [i],
    "dgmr_get_deterministic": [],
    "dgmr_wgrad_dot_floats": [i],
    "dgmr_nonfinite_count": [P, L, P, P],
    "dgmr_debug_flags": [i],
    "dgmr_get_precision": [],
    "dgmr_profile_enable": [i],
    "dgmr_conv_tune": [i, i, i, i],
    "dgmr_profile_variants": [],
    "dgmr_profile_collect": [P, P, P, i],
    "dgmr_profile_collect2": [P, P, P, P, i],
    "dgmr_profile_collect_detail": [P, i],
}
del i, f, L
# buffer sizes: these return int64_t
SIGNATURES_I64 = {"dgmr_reduce_doubles": [c_int, c_int64, c_int], "dgmr_grid_cell_acc_doubles": [c_int64]}

_lib = None


def load():
    """Load libdgmr_hip.so once; raise loudly if it is missing (no CPU / torch fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the DGMR HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.dgmr_abi_version.restype = c_int
    lib.dgmr_abi_version.argtypes = []
    lib.dgmr_last_error.restype = c_char_p
    lib.dgmr_last_error.argtypes = []
    got = lib.dgmr_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"libdgmr_hip.so ABI {got} != expected {ABI_VERSION}; rebuild")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = c_int
        fn.argtypes = argtypes
    lib.dgmr_profile_variant_name.restype = c_char_p
    lib.dgmr_profile_variant_name.argtypes = [c_int]
    for name, argtypes in SIGNATURES_I64.items():
        fn = getattr(lib, name)
        fn.restype = c_int64
        fn.argtypes = argtypes
    lib.dgmr_set_deterministic(int(DETERMINISTIC_DEFAULT))
    _lib = lib
    return lib


def check(rc: int, name: str):
    if rc != 0:
        msg = _lib.dgmr_last_error().decode() if _lib is not None else "?"
        raise RuntimeError(f"{name} failed ({rc}): {msg}")


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        check(rc, name)
