"""ctypes binding of libunet_hip.so (the C ABI declared in include/unet_hip.h).

The library is built in-tree by `build()` (hipcc, --offload-arch=gfx950) and is
the ONLY compute path of this package: there is no CPU or eager-PyTorch
fallback.  `lib()` raises if the shared object is missing.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# UNET_HIP_LIB selects an alternative in-tree build (A/B experiments of kernel variants)
LIB_PATH = os.environ.get("UNET_HIP_LIB") or os.path.join(_HERE, "libunet_hip.so")
CSRC = os.path.join(_HERE, "csrc")

ABI_VERSION = 8      # UNET_ABI_VERSION of include/unet_hip.h this binding was written against

_c = ctypes
_p = _c.c_void_p
_i = _c.c_int
_f = _c.c_float
_sz = _c.c_size_t
_i64 = _c.c_int64



class ActSrc(ctypes.Structure):
    """`unet_act_src` of include/unet_hip.h: an operand activated on load."""
    _fields_ = [("x", _p), ("C", _i), ("alpha", _p), ("beta", _p)]


_ps = _c.POINTER(ActSrc)


class BwdStats(ctypes.Structure):
    """`unet_bwd_stats` of include/unet_hip.h: the layer whose InstanceNorm-backward reductions a
    data-gradient epilogue emits."""
    _fields_ = [("y", _p), ("mean", _p), ("rstd", _p), ("gamma", _p), ("beta", _p), ("mask", _p),
                ("slope", _f), ("partial", _p), ("partial_bytes", _sz), ("tiles_out", _i)]


_pbs = _c.POINTER(BwdStats)

# name -> (restype, argtypes); mirrors include/unet_hip.h one to one
SIGNATURES = {
    "unet_last_error": (_c.c_char_p, []),
    "unet_abi_version": (_i, []),
    "unet_device_count": (_i, []),
    "unet_debug_set_chunk_limit": (_i, [_i64]),
    "unet_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "unet_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "unet_pack_conv3x3_weights": (_i, [_p, _p, _p, _i, _i, _p]),
    "unet_conv3x3_fwd": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_data": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_fwd_bf16": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_data_bf16": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "unet_conv1x1_fwd": (_i, [_p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "unet_conv1x1_bwd_data": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_conv1x1_bwd_weight": (_i, [_p, _i, _p, _p, _i, _i, _p, _sz, _i, _i, _i, _i, _p]),
    "unet_transpose2d": (_i, [_p, _p, _i, _i, _p]),
    "unet_conv3x3_bwd_weight_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "unet_conv3x3_bwd_weight": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _sz, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_weight_bf16": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _sz, _i, _i, _i, _i, _i,
                                          _p]),
    "unet_pack_conv3x3_weights_bf16x3": (_i, [_p, _p, _p, _i, _i, _p]),
    "unet_pack_conv3x3_weights_batched": (_i, [_p, _i, _i, _p]),
    "unet_conv3x3_fwd_bf16x3": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_data_bf16x3": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_weight_bf16x3": (_i, [_p, _i, _p, _p, _i, _i, _p, _p, _sz, _i, _i, _i, _i, _i,
                                          _p]),
    "unet_instnorm_workspace_bytes": (_sz, [_i, _i, _i]),
    "unet_instnorm_stats": (_i, [_p, _p, _p, _f, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _p]),
    "unet_instnorm_lrelu_drop_fwd": (_i, [_p, _p, _p, _p, _f, _p, _i, _i, _i, _p]),
    "unet_instnorm_lrelu_drop_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _sz,
                                          _i, _i, _i, _p]),
    "unet_upsample2x_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "unet_resize_bilinear_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_resize_bilinear_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_upsample2x_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "unet_head1x1_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "unet_head1x1_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "unet_head1x1_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _i, _p]),
    "unet_dice_wce_loss_workspace_bytes": (_sz, [_i, _i, _i]),
    "unet_dice_wce_loss_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _sz, _i, _i, _i, _f, _f, _f, _i, _i,
                                        _p, _f, _p]),
    "unet_dice_wce_loss_shard_stats": (_i, [_p, _p, _p, _p, _sz, _i, _i, _i, _f, _i, _p]),
    "unet_dice_wce_loss_shard_apply": (_i, [_p, _p, _p, _i, _p, _p, _p, _sz, _i, _i, _i, _f, _f, _f,
                                            _i, _i, _p, _f, _p]),
    "unet_wgrad_defer_begin": (_i, []),
    "unet_wgrad_defer_pending": (_i, []),
    "unet_wgrad_defer_flush": (_i, [_p]),
    "unet_wgrad_defer_end": (_i, [_p]),
    "unet_dice_wce_loss_grad": (_i, [_p, _p, _p, _sz, _p, _p, _i, _i, _i, _i, _p]),
    "unet_argmax_dice_counts": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "unet_preprocess_u8": (_i, [_p, _p, _p, _p, _i, _i, _i, _c.POINTER(_f), _c.POINTER(_f), _p]),
    "unet_sgd_nesterov_step": (_i, [_p, _p, _p, _i64, _f, _f, _f, _i, _f, _p]),
    "unet_sgd_nesterov_step_dev": (_i, [_p, _p, _p, _i64, _p, _i, _p]),
    "unet_add_inplace": (_i, [_p, _p, _i64, _p]),
    "unet_conv_in_fwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "unet_conv_in_fwd": (_i, [_ps, _ps, _f, _p, _p, _i, _i, _p, _p, _sz, _c.POINTER(_i), _i, _i, _i,
                              _i, _p]),
    "unet_conv_in_fwd_bf16x3": (_i, [_ps, _ps, _f, _p, _p, _p, _i, _i, _p, _p, _sz, _c.POINTER(_i), _i,
                                     _i, _i, _i, _p]),
    "unet_conv_up_in_fwd_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv_up_in_fwd": (_i, [_ps, _ps, _f, _p, _p, _p, _p, _sz, _c.POINTER(_i), _i, _i, _i, _i,
                                 _p]),
    "unet_conv_in_stats_finalize": (_i, [_p, _p, _sz, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i,
                                         _p]),
    "unet_stem_u8_fwd": (_i, [_p, _c.POINTER(_f), _c.POINTER(_f), _p, _p, _p, _p, _sz,
                              _c.POINTER(_i), _i, _i, _i, _i, _p]),
    "unet_stem_u8_bwd_weight": (_i, [_p, _c.POINTER(_f), _c.POINTER(_f), _p, _p, _p, _sz, _i, _i,
                                     _i, _i, _p]),
    "unet_conv_in_bwd_weight": (_i, [_ps, _f, _p, _p, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _i, _p]),
    "unet_conv_in_bwd_weight_dz_supported": (_i, [_i, _i, _i, _i, _i]),
    "unet_conv_in_bwd_weight_dz": (_i, [_ps, _f, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i,
                                        _p, _sz, _i, _i, _i, _i, _p]),
    "unet_conv_in_bwd_weight_bf16x3": (_i, [_ps, _f, _p, _p, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _i,
                                            _p]),
    "unet_upsample2x_in_fwd": (_i, [_ps, _f, _p, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_data_bs": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_conv_wino_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_set_c32_winograd": (_i, [_i]),
    "unet_conv_c32_is_winograd": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv_up_c32_is_winograd": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_wino_weight_floats": (_sz, [_i, _i]),
    "unet_pack_wino_weights": (_i, [_p, _p, _p, _i, _i, _p]),
    "unet_pack_wino_weights_batched": (_i, [_p, _i, _i, _p]),
    "unet_conv_in_fwd_wino": (_i, [_ps, _ps, _f, _p, _p, _p, _p, _sz, _c.POINTER(_i), _i, _i, _i,
                                   _i, _p]),
    "unet_conv3x3_bwd_weight_is_winograd": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv_up_wino_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv_up_in_fwd_wino": (_i, [_ps, _ps, _f, _p, _p, _p, _p, _sz, _c.POINTER(_i), _i, _i,
                                      _i, _i, _p]),
    "unet_conv3x3_bwd_data_bs_wino": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_instnorm_bwd_coefs": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "unet_conv3x3_bwd_data_dz_wino": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i,
                                           _p, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_conv3x3_bwd_data_bs_b16": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_conv3x3_bwd_data_bs_b16_wb": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _pbs,
                                             _p]),
    "unet_conv3x3_bwd_data_bs_bf16x3": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _pbs,
                                             _p]),
    "unet_conv3x3_up_bwd_data_bs": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_instnorm_lrelu_drop_bwd_partials": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p,
                                                   _p, _i, _p, _sz, _i, _i, _i, _p]),
    "unet_instnorm_lrelu_drop_bwd_partials_b16": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p,
                                                       _p, _i, _p, _sz, _i, _i, _i, _p]),
    "unet_conv_in_fwd_b16": (_i, [_ps, _ps, _f, _p, _p, _i, _i, _p, _p, _sz, _c.POINTER(_i), _i, _i,
                                  _i, _i, _p]),
    "unet_conv_in_fwd_b16_wb": (_i, [_ps, _ps, _f, _p, _p, _p, _i, _i, _p, _p, _sz, _c.POINTER(_i),
                                     _i, _i, _i, _i, _p]),
    "unet_conv_in_stats_finalize_b16": (_i, [_p, _p, _sz, _i, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i,
                                             _i, _p]),
    "unet_conv_in_bwd_weight_b16": (_i, [_ps, _f, _p, _p, _i, _i, _i, _i, _p, _sz, _i, _i, _i, _i,
                                         _p]),
    "unet_conv3x3_bwd_data_b16": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "unet_instnorm_lrelu_drop_bwd_b16": (_i, [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                              _sz, _i, _i, _i, _p]),
    "unet_upsample2x_in_fwd_b16": (_i, [_ps, _f, _p, _i, _i, _i, _p]),
    "unet_upsample2x_bwd_taps_b16": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "unet_conv3x3_up_bwd_weight_b16": (_i, [_ps, _f, _p, _p, _i, _i, _p, _sz, _i, _i, _i, _i, _p]),
    "unet_conv_up_in_fwd_b16_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "unet_conv_up_in_fwd_b16": (_i, [_ps, _ps, _f, _p, _p, _p, _p, _p, _sz, _c.POINTER(_i), _i, _i, _i,
                                     _i, _p]),
    "unet_conv3x3_up_bwd_data_b16": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_conv3x3_up_bwd_data_bs_b16": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _pbs, _p]),
    "unet_conv3x3_up_bwd_data_bs_b16_wb": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _pbs,
                                           _p]),
    "unet_head1x1_in_fwd_b16": (_i, [_ps, _f, _p, _p, _p, _i, _i, _i, _p]),
    "unet_head1x1_in_bwd_b16": (_i, [_ps, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _p]),
    "unet_upsample2x_bwd_taps": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "unet_conv3x3_up_bwd_weight_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "unet_conv3x3_up_bwd_weight": (_i, [_ps, _f, _p, _p, _i, _i, _p, _sz, _i, _i, _i, _i, _p]),
    "unet_conv3x3_up_bwd_data": (_i, [_p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "unet_head1x1_in_fwd": (_i, [_ps, _f, _p, _p, _p, _i, _i, _i, _p]),
    "unet_head1x1_in_bwd": (_i, [_ps, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _p]),
    "unet_head1x1_in_bwd_bs": (_i, [_ps, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _pbs, _p]),
    "unet_head1x1_in_bwd_bs_b16": (_i, [_ps, _f, _p, _p, _p, _p, _p, _p, _sz, _i, _i, _i, _pbs, _p]),
}

_lib = None


class UNetHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile libunet_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    jobs = str(min(8, os.cpu_count() or 1))
    proc = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
        print(proc.stderr)
    if proc.returncode != 0:
        raise UNetHipError("building libunet_hip.so failed (see output above)")
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UNetHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU / eager fallback for the HIP path)")
    # torch must be imported first so its bundled libamdhip64.so.7 (same soname) is the
    # one HIP runtime of the process: streams and device pointers are then shared.
    import torch  # noqa: F401
    handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if handle.unet_abi_version() != ABI_VERSION:
        raise UNetHipError("libunet_hip.so ABI version mismatch; rebuild")
    _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().unet_last_error()
        raise UNetHipError(f"libunet_hip error {rc}: {msg.decode() if msg else '?'}")
