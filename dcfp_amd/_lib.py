"""ctypes binding of libdcfp_hip.so (the C-ABI declared in include/dcfp_hip.h).

The library is the only compute path of this package: there is no CPU or eager-PyTorch
fallback.  `lib()` raises if the shared object is missing; every wrapper raises
RuntimeError on a non-zero status (<0 descriptor error, >0 hipError_t).
"""
import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCFP_LIB") or os.path.join(_HERE, "libdcfp_hip.so")   # DCFP_LIB: A/B builds
CSRC = os.path.join(_HERE, "csrc")

_lib = None
_lock = threading.Lock()


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Cin", "H", "W", "Cout", "KH", "KW", "stride", "pad", "dil", "Hout", "Wout",
                 "x_pitch", "dy_pitch")]


class EicEntry(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("grad", C.c_void_p), ("eic", C.c_void_p),
                ("n", C.c_int32), ("pad_", C.c_int32)]


class SgdEntry(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("momentum_buf", C.c_void_p),
                ("n", C.c_int64), ("first_chunk", C.c_int64),
                ("weight_decay", C.c_float), ("pad_", C.c_int32)]


class BnRunning(C.Structure):
    """DcfpBnRunning: nn.BatchNorm2d's training-mode bookkeeping folded into the statistics kernels."""
    _fields_ = [("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p),
                ("momentum", C.c_float), ("pad_", C.c_int32)]


class WpEntry(C.Structure):
    """DcfpWpEntry: one weight tensor -> permuted copy of the multi-tensor refresh."""
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("first_block", C.c_int64), ("n_blocks", C.c_int64)] + \
               [(n, C.c_int32) for n in ("T", "Ck", "CkP", "M", "Mpad", "sAm", "sAc", "perm8")]


SGD_CHUNK = 16384
CONV_FWD, CONV_DGRAD, CONV_WGRAD = 0, 1, 2
E_BADDESC, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3   # DCFP_E_* of include/dcfp_hip.h

_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_D = C.POINTER(ConvDesc)
_R = C.POINTER(BnRunning)

# name -> (restype, argtypes); mirrors include/dcfp_hip.h one to one
SIGNATURES = {
    "dcfp_abi_version": (_I, []),
    "dcfp_conv2d_workspace_bytes": (_Z, [_D, _I]),
    "dcfp_conv2d_wp_layout": (_I, [_D, _I, C.POINTER(WpEntry)]),
    "dcfp_conv2d_permute_weights_multi_f32": (_I, [_P, _I, _L, _P]),
    "dcfp_conv2d_pitch_supported": (_I, [_D]),
    "dcfp_conv2d_kernel_name": (_I, [_D, _I, C.c_char_p, _I]),
    "dcfp_conv2d_executed_fraction": (C.c_double, [_D, _I]),
    "dcfp_conv2d_workspace_is_scratch": (_I, [_D, _I]),
    "dcfp_conv2d_dgrad_fanin_supported": (_I, [_D]),
    "dcfp_conv2d_dgrad_fanin_f32_nchw": (_I, [_D, _P, _L, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "dcfp_conv2d_dgrad_fanin_red_slots": (_L, [_D]),
    "dcfp_conv2d_dgrad_fanin_red_f32_nchw": (_I, [_D, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "dcfp_bn_bwd_sums_from_partials_f32": (_I, [_P, _L, _I, _P, _F, _P, _P, _P, _P, _P]),
    "dcfp_conv2d_xform_bytes": (_Z, [_D]),
    "dcfp_conv2d_fwd_keep_f32_nchw": (_I, [_D, _P, _P, _P, _L, _P, _Z, _P, _P, _Z, _P]),
    "dcfp_conv2d_wgrad_kept_f32_nchw": (_I, [_D, _P, _L, _P, _Z, _P, _P, _Z, _P]),
    "dcfp_conv2d_fwd_f32_nchw": (_I, [_D, _P, _P, _P, _P, _L, _P, _Z, _I, _P]),
    "dcfp_conv2d_fwd_fused_f32_nchw": (_I, [_D, _P, _P, _P, _P, _P, _I, _P, _P, _Z, _I, _P]),
    "dcfp_conv2d_dgrad_f32_nchw": (_I, [_D, _P, _L, _P, _P, _I, _P, _Z, _I, _P]),
    "dcfp_conv2d_wgrad_f32_nchw": (_I, [_D, _P, _L, _P, _P, _P, _P, _Z, _P]),
    "dcfp_bn_workspace_bytes": (_Z, [_I, _I, _I]),
    "dcfp_bn_stats_f32": (_I, [_P, _L, _I, _I, _I, _P, _P, _R, _P, _Z, _P]),
    "dcfp_bn_apply_f32": (_I, [_P, _P, _P, _P, _P, _F, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P]),
    "dcfp_bn_bwd_reduce_f32": (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _F, _I, _I, _I, _I, _P, _P, _P, _P,
                                    _P, _Z, _P]),
    "dcfp_bn_update_running_f32": (_I, [_P, _P, _I, _F, _F, _P, _P, _P, _P]),
    "dcfp_syncbn_combine_f32": (_I, [_P, _I, _I, _P, _P, _P, _R, _P]),
    "dcfp_syncbn_p2p_mailbox_bytes": (_Z, [_I, _I]),
    "dcfp_p2p_alloc": (_I, [_Z, _I, C.POINTER(C.c_void_p)]),
    "dcfp_p2p_free": (_I, [_P]),
    "dcfp_p2p_export": (_I, [_P, _P]),
    "dcfp_p2p_import": (_I, [_P, C.POINTER(C.c_void_p)]),
    "dcfp_p2p_unmap": (_I, [_P]),
    "dcfp_syncbn_p2p_exchange_f32": (_I, [C.POINTER(C.c_void_p), _I, _I, C.c_uint32, _I, _P, _I, _I, _P, _R,
                                          C.c_uint32, _P, _P]),
    "dcfp_bn_apply_relu_mask_f32": (_I, [_P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _P]),
    "dcfp_conv2d_fwd_stat_slots": (_L, [_D, _P, _L]),
    "dcfp_conv2d_fwd_stats_f32_nchw": (_I, [_D, _P, _P, _P, _L, _P, _P, _Z, _I, _P]),
    "dcfp_bn_stats_from_partials_f32": (_I, [_P, _L, _I, _I, _P, _P, _R, _P]),
    "dcfp_bn_bwd_apply_f32": (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _F, _P, _P, _F, _P, _I, _P, _P,
                                   _I, _I, _I, _I, _I, _P]),
    "dcfp_bn_bwd_fused_sync_bytes": (_Z, [_I, _I, _I]),
    "dcfp_bn_bwd_fused_f32": (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _F, _F, _I, _P, _P, _I, _I, _I, _I, _I,
                                   _P, _P, _P, _P, _P, _Z, C.c_uint32, C.c_uint32, _P, _P]),
    "dcfp_maxpool3x3s2_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dcfp_maxpool3x3s2_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dcfp_rowsum_f32": (_I, [_P, _L, _P, _F, _I, _I, _I, _P]),
    "dcfp_broadcast_hw_f32": (_I, [_P, _F, _P, _L, _I, _I, _I, _I, _P]),
    "dcfp_add_f32": (_I, [_P, _P, _P, _L, _P]),
    "dcfp_channel_scale_f32": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "dcfp_upsample_bilinear_fwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dcfp_upsample_bilinear_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dcfp_upsample_ce_workspace_bytes": (_Z, [_I, _I, _I]),
    "dcfp_upsample_ce_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P,
                                      _Z, _P]),
    "dcfp_upsample_ce_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dcfp_ohem_zoom_gt_prob_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dcfp_ohem_threshold_f32": (_I, [_P, _P, _L, _I, _F, _L, _P, _P]),
    "dcfp_ohem_keep_mask_u8": (_I, [_P, _P, _L, _P, _P]),
    "dcfp_upsample_margin_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dcfp_maxfilter2d_s1_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "dcfp_upsample_wce_workspace_bytes": (_Z, [_I, _I, _I]),
    "dcfp_upsample_wce_fwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _Z, _P]),
    "dcfp_upsample_wce_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dcfp_upsample_argmax_f32": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "dcfp_confusion_matrix_i64": (_I, [_P, _P, _I, _L, _I, _P, _P]),
    "dcfp_eic_update_f32": (_I, [_P, _I, _F, _F, _P]),
    "dcfp_sgd_momentum_f32": (_I, [_P, _I, _L, _F, _F, _I, _P]),
}


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 into libdcfp_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libdcfp_hip.so failed")
    return LIB_PATH


def lib():
    """Load the shared object (after torch, so that it binds to the HIP runtime already in
    the process) and attach signatures.  No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C dcfp_amd/csrc`). dcfp_amd has no non-HIP fallback.")
        try:
            import torch  # noqa: F401  (loads libamdhip64.so.7 first; ours resolves to it)
        except Exception:  # pragma: no cover
            pass
        handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        kind = "descriptor/shape" if status < 0 else "hipError_t"
        raise RuntimeError(f"{what} failed: status {status} ({kind})")
