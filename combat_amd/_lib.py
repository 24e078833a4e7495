"""ctypes binding of libcombat_hip.so (the C ABI declared in include/combat_hip.h).

The product path has no fallback: if the shared library is missing or does not export a symbol
the header declares, importing this module raises.  Build it with ``python -m combat_amd.build``
(hipcc cross-compiles gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COMBAT_HIP_LIB") or os.path.join(HERE, "libcombat_hip.so")   # override: profiling builds

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    """struct combat_conv_args"""

    _fields_ = [
        ("N", c_i32), ("H", c_i32), ("W", c_i32), ("C", c_i32),
        ("P", c_i32), ("Q", c_i32), ("K", c_i32),
        ("R", c_i32), ("S", c_i32), ("stride", c_i32), ("pad", c_i32),
        ("mode", c_i32),
        ("src", c_vp), ("wpack", c_vp), ("kpad", c_i32), ("rows_pad", c_i32), ("dst", c_vp),
        ("pro_scale", c_vp), ("pro_shift", c_vp), ("pro_group_stride", c_i32), ("pro_act", c_i32),
        ("pro_slope", c_f32),
        ("bias", c_vp), ("add_pre", c_vp), ("mask_x", c_vp), ("mask_scale", c_vp), ("mask_shift", c_vp),
        ("mask_activated", c_i32), ("mask_group_stride", c_i32), ("mask_slope", c_f32), ("mask_mul_scale", c_i32),
        ("add_post", c_vp), ("tanh_out", c_i32),
        ("stats_kind", c_i32), ("stats", c_vp), ("xh_mean", c_vp), ("xh_rstd", c_vp),
        ("act_dst", c_vp), ("act_scale", c_vp), ("act_shift", c_vp), ("act_slope", c_f32),
        ("workspace", c_vp), ("workspace_bytes", c_i64),
        ("tile", c_i32),
        ("src2", c_vp), ("wpack2", c_vp), ("kpad2", c_i32), ("rows_pad2", c_i32),
        ("pro_act_dst", c_vp),
    ]


class WgradArgs(C.Structure):
    """struct combat_wgrad_args"""

    _fields_ = [
        ("N", c_i32), ("H", c_i32), ("W", c_i32), ("C", c_i32),
        ("P", c_i32), ("Q", c_i32), ("K", c_i32),
        ("R", c_i32), ("S", c_i32), ("stride", c_i32), ("pad", c_i32),
        ("src", c_vp), ("dy", c_vp), ("dw", c_vp), ("k_real", c_i32), ("c_real", c_i32),
        ("pro_scale", c_vp), ("pro_shift", c_vp), ("pro_group_stride", c_i32), ("pro_act", c_i32),
        ("pro_slope", c_f32), ("split", c_i32),
        ("workspace", c_vp), ("workspace_bytes", c_i64),
        ("defer_reduce", c_i32), ("reserved", c_i32),
        ("reduce_first", c_vp),
    ]


class PackDesc(C.Structure):
    """struct combat_pack_desc"""

    _fields_ = [("w", c_vp), ("wf", c_vp), ("wd", c_vp),
                ("K", c_i32), ("taps", c_i32), ("c_real", c_i32), ("C", c_i32), ("dup_hilo", c_i32),
                ("rows_pad_f", c_i32), ("kpad_f", c_i32), ("rows_pad_d", c_i32), ("kpad_d", c_i32), ("reserved", c_i32),
                ("row_scale", c_vp)]


class BnDesc(C.Structure):
    """struct combat_bn_desc"""

    _fields_ = [("gamma", c_vp), ("beta", c_vp), ("running_mean", c_vp), ("running_var", c_vp),
                ("scale", c_vp), ("shift", c_vp), ("C", c_i32), ("reserved", c_i32)]


# name -> (restype, argtypes); one entry per function declared in include/combat_hip.h
SIGNATURES = {
    "combat_version": (C.c_char_p, []),
    "combat_abi_version": (C.c_int, []),
    "combat_set_deterministic": (None, [C.c_int]),
    "combat_get_deterministic": (C.c_int, []),
    "combat_conv_gemm": (C.c_int, [C.POINTER(ConvArgs), c_vp]),
    "combat_conv_gemm_pair": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(ConvArgs), c_vp]),
    "combat_conv_workspace_bytes": (c_i64, [C.POINTER(ConvArgs)]),
    "combat_conv_pick_tile": (C.c_int, [C.POINTER(ConvArgs)]),
    "combat_conv_stats_granule": (C.c_int, [C.c_int]),
    "combat_conv_stats_layout": (C.c_int, [C.POINTER(ConvArgs), C.POINTER(c_i32), C.POINTER(c_i32)]),
    "combat_conv_wgrad": (C.c_int, [C.POINTER(WgradArgs), c_vp]),
    "combat_conv_wgrad_workspace_bytes": (c_i64, [C.POINTER(WgradArgs)]),
    "combat_conv_wgrad_reduce": (C.c_int, [C.POINTER(WgradArgs), c_vp]),
    "combat_pack_weights": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32,
                                      c_i32, c_vp]),
    "combat_norm_finalize": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp]),
    "combat_norm_scratch_bytes": (c_i64, [c_i32, c_i32]),
    "combat_bn_eval_fold": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp, c_vp]),
    "combat_pack_weights_batch": (C.c_int, [c_vp, c_i32, c_vp]),
    "combat_bn_eval_fold_batch": (C.c_int, [c_vp, c_i32, c_f32, c_vp]),
    "combat_group_stats": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_norm_bwd_finalize": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_i64, c_vp]),
    "combat_norm_bwd_apply": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "combat_norm_act_fused": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "combat_norm_add_act_fused": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "combat_norm_bwd_fused": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_vp, c_i64, c_vp, c_vp]),
    "combat_group_stats_bwd": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "combat_unet_up_fused": (C.c_int, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp,
                                       c_vp, c_vp, c_vp, c_vp]),
    "combat_unet_up_bwd_fused": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "combat_unet_up_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_unet_up_bwd": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_trigger_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "combat_trigger_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_f32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_i32, c_vp, c_vp]),
    "combat_augment_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "combat_augment_bwd": (C.c_int, [c_vp, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "combat_head_fwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp]),
    "combat_head_bwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                  c_vp]),
    "combat_head_fwd_bwd": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp,
                                      c_vp, c_vp, c_vp, c_vp, c_vp]),
    "combat_head_bwd_weights": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "combat_comm_unique_id": (C.c_int, [c_vp]),
    "combat_comm_init_rank": (C.c_int, [C.POINTER(c_vp), c_i32, c_vp, c_i32]),
    "combat_comm_destroy": (C.c_int, [c_vp]),
    "combat_allreduce": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "combat_sgd_nesterov": (C.c_int, [c_vp, c_vp, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "combat_image_to_c8": (C.c_int, [c_vp, c_i32, c_i32, c_vp, c_vp]),
    "combat_nhwc_to_nchw_f32": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_nchw_to_nhwc_bf16": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_memset_zero": (C.c_int, [c_vp, c_i64, c_vp]),
    "combat_copy3": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "combat_relu_mask": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp]),
    "combat_colsum": (C.c_int, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "combat_log_terms": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "combat_maxpool2": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_elu_affine": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "combat_affine_act": (C.c_int, [c_vp, c_i64, c_i32, c_vp, c_vp, c_i64, c_f32, c_vp, c_vp]),
    "combat_dct_u8": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "combat_linear_nhwc": (C.c_int, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp]),
    "combat_grid_head_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "combat_wanet_grid": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp]),
    "combat_warp_fwd": (C.c_int, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_warp_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_warp_bwd_input": (C.c_int, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "combat_wanet_field_bwd": (C.c_int, [c_vp, c_i32, c_vp, c_vp, c_i32, c_i32, c_f32, c_f32, c_vp, c_vp, c_vp, c_i32, c_vp,
                                         c_vp, c_vp, c_vp, c_vp]),
    "combat_plan_create": (c_vp, []),
    "combat_plan_destroy": (None, [c_vp]),
    "combat_plan_record": (C.c_int, [c_vp, c_i32]),
    "combat_plan_record_cancel": (C.c_int, []),
    "combat_plan_set_after": (C.c_int, [c_vp, c_i32]),
    "combat_plan_size": (c_i32, [c_vp]),
    "combat_plan_run": (C.c_int, [c_vp, c_i32, c_i32, c_vp, C.POINTER(c_vp), c_i32]),
    "combat_plan_join": (C.c_int, [c_vp, c_vp, C.POINTER(c_vp), c_i32]),
    "combat_plan_failed_call": (c_i32, [c_vp]),
}

TILE_128x128, TILE_128x64, TILE_64x64, TILE_128x16, TILE_64x128 = 1, 2, 3, 4, 5
TILE_H256x64, TILE_H128x128, TILE_H128x64, TILE_H64x64, TILE_D128x64, TILE_D128x32 = 6, 7, 8, 9, 10, 11
TILE_G128x64, TILE_G128x32 = 12, 13
TILE_D256x64, TILE_C8, TILE_D256W64, TILE_S128x64, TILE_K8 = 14, 15, 16, 17, 18
STATS_PER_WORKGROUP = 4     # | into ConvArgs.stats_kind: one statistics row per workgroup (include/combat_hip.h)


class CombatHipError(RuntimeError):
    pass


def load(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            "combat_amd: %s not found -- the HIP library is required (no CPU fallback exists); "
            "build it with `python -m combat_amd.build`" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("combat_amd: %s does not export %s" % (path, name)) from e
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(status: int, what: str, detail: str = "") -> None:
    """Map a C status code to a Python exception naming the kernel and the shape."""
    if status == 0:
        return
    kind = {-1: "invalid shape/argument", -2: "HIP launch failed"}.get(status, "status %d" % status)
    raise CombatHipError("%s: %s %s" % (what, kind, detail))
