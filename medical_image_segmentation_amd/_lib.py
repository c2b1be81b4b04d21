"""ctypes binding of libmisamd.so (the C-ABI declared in include/misamd.h).

The shared library is built in-tree by `make -C medical_image_segmentation_amd/csrc`
(or `__graft_entry__.build()`).  There is NO fallback: if the library is missing or a
call fails, a RuntimeError is raised — the reference signals errors by exceptions too
(e.g. train.py:211).
"""
import ctypes as C
import os

# torch ships its own HIP runtime (libamdhip64): importing it first makes libmisamd.so bind
# to that same runtime instance, so both see the same devices, streams and allocations.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISAMD_LIBRARY") or os.path.join(_HERE, "libmisamd.so")

F32, BF16 = 0, 1

_vp, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/misamd.h one to one
PROTOTYPES = {
    "mis_version": (_i, []),
    "mis_arch": (C.c_char_p, []),
    "mis_last_error": (C.c_char_p, []),
    "mis_loss_metrics_workspace_bytes": (_sz, [_i, _i, _i]),
    "mis_loss_metrics_fwd": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_loss_metrics_fwd_static": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "mis_dice_from_sums": (_i, [_vp, _i, _f, _i, _vp, _vp]),
    "mis_loss_bwd": (_i, [_vp, _i, _i64, _i64, _i64, _vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    "mis_maxpool2x2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mis_maxpool2x2_bwd_workspace_bytes": (_sz, [_i]),
    "mis_maxpool2x2_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_upsample2x_cat_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "mis_upsample2x_cat_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mis_dropout2d_noise": (_i, [C.c_uint64, C.c_uint64, _i, _i, _f, _vp, _vp]),
    "mis_conv3x3_packed_weight_bytes": (_sz, [_i, _i, _i, _i]),
    "mis_pack_conv3x3_weights": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "mis_set_conv_path": (_i, [_i]),
    "mis_set_conv_launch": (_i, [_i, _i]),
    "mis_conv3x3_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mis_conv3x3_stats_rows": (_i, [_i, _i, _i, _i, _i, _i]),
    "mis_pack_conv3x3_job": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "mis_pack_conv3x3_weights_multi": (_i, [_vp, _i, _i, _i, _vp]),
    "mis_pack_conv3x3_weights_pair": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "mis_conv3x3_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_conv3x3_dgrad_bn_sums_supported": (_i, [_i] * 6),
    "mis_conv3x3_dgrad_bn_sums": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_conv3x3_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mis_conv3x3_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "mis_conv3x3_wgrad_slabs_supported": (_i, [_i] * 8),
    "mis_conv3x3_wgrad_slabs": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mis_wgrad_reduce_multi": (_i, [_vp, _i, _vp]),
    "mis_conv3x3_dilated_supported": (_i, [_i] * 7),
    "mis_conv3x3_dilated_stats_rows": (_i, [_i] * 7),
    "mis_conv3x3_dilated_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_conv3x3_dilated_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "mis_clip_adamw_workspace_bytes": (_sz, [_i]),
    "mis_grad_sqsum": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mis_clip_adamw_step": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _d, _d, _d, _d, _d, _i64, _i, _vp, _vp]),
    "mis_adamw_step_constants": (None, [_d, _d, _d, _d, _i64, _vp]),
    "mis_clip_adamw_step_dyn": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _f, _d, _d, _d, _vp, _i, _vp, _vp]),
    "mis_conv3x3_cat_supported": (_i, [_i] * 7),
    "mis_conv3x3_cat_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_conv3x3_split_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_conv3x3_cat_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "mis_bn_finalize": (_i, [_vp, _i, _i, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_eval_affine": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    "mis_bn_relu_apply": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp]),
    "mis_bn_relu_bwd_workspace_bytes": (_sz, [_i64, _i]),
    "mis_bn_relu_bwd_reduce": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_relu_bwd_dz_rows": (_i, [_i, _i64, _i]),
    "mis_bn_relu_bwd_dz_head": (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_relu_bwd_dz": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_relu_bwd_dz_scaled": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_relu_bwd_dz_scaled4": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_bwd_reduce": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_bn_bwd_dz": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_add_bn_relu_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _vp, _vp]),
    "mis_relu_mask_bwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp]),
    "mis_mdam_workspace_bytes": (_sz, [_i, _i]),
    "mis_mdam_window_sums": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mis_mdam_window_sums_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mis_channel_scale": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp]),
    "mis_channel_dot": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp]),
    "mis_channel_sum_workspace_bytes": (_sz, [_i64, _i]),
    "mis_channel_sum": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp]),
    "mis_conv1x1_fwd": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_conv1x1_bwd_workspace_bytes": (_sz, [_i64, _i, _i]),
    "mis_conv1x1_bwd": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_nchw_f32_to_nhwc": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mis_maxpool3x3s1_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "mis_maxpool3x3s1_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "mis_set_maxpool3_path": (_i, [_i]),
    "mis_sum_gradients_workspace_bytes": (_sz, [_i]),
    "mis_sum_gradients": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_dwconv_compose_multi": (_i, [_vp, _i, _i, _vp]),
    "mis_dwconv_compose_bwd_multi": (_i, [_vp, _i, _vp]),
    "mis_amsfn_workspace_bytes": (_sz, [_i, _i, _i]),
    "mis_amsfn_lazy_workspace_bytes": (_sz, [_i, _i, _i]),
    "mis_amsfn_fuse_bwd_lazy": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp]),
    "mis_amsfn_fuse_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mis_amsfn_fuse_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp]),
    "mis_dwconv_compose_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "mis_dwconv_compose_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "mis_slide_gather_tiles": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp]),
    "mis_argmax_classes": (_i, [_vp, _i, _i64, _i64, _i64, _i, _i, _i, _vp, _vp]),
    "mis_slide_blend": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
}



class PackJob(C.Structure):
    """mis_pack_job of include/misamd.h."""
    _fields_ = [("w", _vp), ("out_fwd", _vp), ("out_t", _vp), ("total_fwd", _i64), ("total_t", _i64),
                ("Cw_out", _i), ("Cw_in", _i), ("cpg_fwd", _i), ("nt_fwd", _i), ("cpg_t", _i), ("nt_t", _i),
                ("first_block", _i), ("n_blocks", _i)]


class ComposeJob(C.Structure):
    """mis_compose_job of include/misamd.h."""
    _fields_ = [("P", _vp), ("D", _vp), ("bd", _vp), ("bp", _vp), ("W", _vp), ("b", _vp),
                ("Cout", _i), ("Cin", _i), ("first_block", _i), ("n_blocks", _i)]


class ComposeBwdJob(C.Structure):
    """mis_compose_bwd_job of include/misamd.h."""
    _fields_ = [("P", _vp), ("D", _vp), ("bd", _vp), ("dW", _vp), ("db", _vp), ("dP", _vp), ("dD", _vp), ("dbd", _vp),
                ("Cout", _i), ("Cin", _i)]


_lib = None
PATH_EPOCH = 0        # bumped by every mis_set_conv_path call: cached operand images depend on the path


def load():
    """Loads the library once; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(there is no CPU or eager fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def call(name, *args):
    """Calls a status-returning entry point; non-zero -> RuntimeError(mis_last_error())."""
    lib = load()
    if name == "mis_set_conv_path":
        global PATH_EPOCH
        PATH_EPOCH += 1
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.mis_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (status {rc}): {msg}")


def query(name, *args):
    return getattr(load(), name)(*args)
