"""ctypes binding of libsqdet_hip.so (include/sqdet.h).  There is NO CPU fallback: if the
library is missing or a call fails this raises -- the product path never routes around the
HIP kernels."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# (SQDET_LIB: another build of the same library -- same-box A/B of a compile-time variant, tools/ab_lib.sh)
LIB_PATH = os.environ.get("SQDET_LIB") or os.path.join(_PKG, "libsqdet_hip.so")

SQDET_OK = 0
SQDET_EUNSUPPORTED = -2
F32, F16 = 0, 1
PAD_SAME, PAD_VALID = 0, 1
ARCH_SQUEEZEDET, ARCH_SQUEEZEDET_PLUS, ARCH_RESNET50 = 0, 1, 2

_lib = None

vp, ci, cf, cd, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes): every symbol include/sqdet.h declares
SIGNATURES = {
    "sqdet_version": (C.c_char_p, []),
    "sqdet_last_error": (C.c_char_p, []),
    "sqdet_set_option": (ci, [C.c_char_p, ci]),
    "sqdet_conv_packed_bytes": (sz, [ci, ci, ci, ci]),
    "sqdet_conv_pack_weights": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "sqdet_conv2d_nhwc_fwd": (ci, [vp, vp, vp, vp] + [ci] * 12 + [vp]),
    "sqdet_conv2d_add_nhwc_fwd": (ci, [vp, vp, vp, vp] + [ci] * 12 + [vp]),
    "sqdet_conv2d_res_nhwc_fwd": (ci, [vp, vp, vp, vp, vp] + [ci] * 12 + [vp]),
    "sqdet_fold_batchnorm": (ci, [vp] * 6 + [cf, vp, vp, ci, ci, ci, vp]),
    "sqdet_fold_batchnorm_bwd_workspace_bytes": (sz, [ci] * 3),
    "sqdet_fold_batchnorm_bwd": (ci, [vp] * 7 + [cf, vp, vp, vp, vp, ci, ci, ci, vp]),
    "sqdet_fold_batchnorm_bwd_many_table_bytes": (sz, [ci]),
    "sqdet_fold_batchnorm_bwd_many_prepare": (ci, [vp] * 14 + [ci, vp, vp, vp]),
    "sqdet_fold_batchnorm_bwd_many": (ci, [vp, ci, ci, ci, cf, vp]),
    "sqdet_subsample_nhwc": (ci, [vp, vp] + [ci] * 6 + [vp]),
    "sqdet_maxpool_nhwc_fwd": (ci, [vp, vp] + [ci] * 8 + [vp]),
    "sqdet_maxpool_nhwc_fwd_idx": (ci, [vp, vp, vp] + [ci] * 8 + [vp]),
    "sqdet_maxpool_nhwc_bwd_idx": (ci, [vp, vp, vp, vp] + [ci] * 9 + [vp]),
    "sqdet_stem_conv_pool_fwd": (ci, [vp, vp, vp, vp] + [ci] * 8 + [vp]),
    "sqdet_stem_conv_pool_squeeze_supported": (ci, [ci] * 9),
    "sqdet_stem_conv_pool_squeeze_fwd": (ci, [vp] * 6 + [ci] * 9 + [vp]),
    "sqdet_fire_fwd": (ci, [vp] * 9 + [ci] * 8 + [vp]),
    "sqdet_fire_fwd_keep": (ci, [vp] * 9 + [ci] * 8 + [vp]),
    "sqdet_fire_maxpool_fwd": (ci, [vp] * 10 + [ci] * 8 + [vp]),
    "sqdet_fire_expand_fwd": (ci, [vp] * 6 + [ci] * 8 + [vp]),
    "sqdet_fire_expand_pair_supported": (ci, [ci] * 7),
    "sqdet_fire_squeeze_next_supported": (ci, [ci] * 6),
    "sqdet_fire_squeeze_next_fwd": (ci, [vp] * 10 + [ci] * 9 + [vp]),
    "sqdet_fire_expand_squeeze_next_supported": (ci, [ci] * 6),
    "sqdet_fire_expand_squeeze_next_fwd": (ci, [vp] * 8 + [ci] * 9 + [vp]),
    "sqdet_fire_chain_stream_bytes": (sz, [ci] * 5),
    "sqdet_fire_chain_pack": (ci, [vp] * 4 + [ci] * 5 + [vp]),
    "sqdet_fire_chain_fwd": (ci, [vp] * 7 + [ci] * 8 + [vp]),
    "sqdet_interpret_output": (ci, [vp] * 7 + [ci] * 5 + [cf, cf, cf, ci, vp]),
    "sqdet_filter_prediction": (ci, [vp] * 8 + [ci] * 5 + [cd, cf, vp]),
    "sqdet_detect_filter": (ci, [vp] * 8 + [ci] * 5 + [cf, cf, cf, ci, ci, cd, ci, vp]),
    "sqdet_detect_filter_scored": (ci, [vp] * 8 + [ci] * 5 + [cf, cf, cf, ci, ci, cd, ci, ci, vp]),
    "sqdet_net_set_signal": (ci, [vp, ci, vp]),
    "sqdet_net_set_post_job": (ci, [vp] * 9 + [ci] * 5 + [cf, cf, cf, ci, ci, cd, ci]),
    "sqdet_net_rider_capacity": (ci, [vp]),
    "sqdet_net_overlap_layer": (ci, [vp]),
    "sqdet_convdet_fwd": (ci, [vp] * 5 + [ci] * 7 + [vp]),
    "sqdet_convdet_scores_supported": (ci, [ci] * 4),
    "sqdet_net_set_scores": (ci, [vp, vp]),
    "sqdet_net_scores_supported": (ci, [vp]),
    "sqdet_conv_pack_weights_bwd_data": (ci, [vp, vp, ci, ci, ci, ci, vp]),
    "sqdet_conv_pack_many_table_bytes": (sz, [ci]),
    "sqdet_conv_pack_many_prepare": (ci, [vp, vp, vp, vp, vp, vp, ci, ci, vp, C.POINTER(ci)]),
    "sqdet_conv_pack_many_prepare_bn": (ci, [vp] * 12 + [cf, ci, ci, vp, vp]),
    "sqdet_conv_pack_many": (ci, [vp, ci, ci, ci, vp]),
    "sqdet_conv2d_nhwc_bwd_data": (ci, [vp, vp, vp] + [ci] * 10 + [vp]),
    "sqdet_conv2d_nhwc_bwd_data_relu": (ci, [vp, vp, vp, vp] + [ci] * 10 + [vp]),
    "sqdet_conv2d_bwd_filter_workspace_bytes": (sz, [ci] * 6),
    "sqdet_conv2d_nhwc_bwd_filter": (ci, [vp, vp, vp, vp, vp, cf, cf, vp] + [ci] * 11 + [vp]),
    "sqdet_conv2d_nhwc_bwd_filter_partial": (ci, [vp, vp, vp] + [ci] * 12 + [vp]),
    "sqdet_slab_reduce_many_table_bytes": (sz, [ci]),
    "sqdet_slab_reduce_many_prepare": (ci, [vp] * 11 + [ci, vp, vp]),
    "sqdet_slab_reduce_many": (ci, [vp, ci, ci, cf, vp]),
    "sqdet_relu_bwd": (ci, [vp, vp, sz, ci, vp]),
    "sqdet_convert_scale": (ci, [vp, ci, vp, ci, cf, sz, vp]),
    "sqdet_scale_mask": (ci, [vp, vp, vp, cf, sz, ci, vp]),
    "sqdet_scale_mask_relu": (ci, [vp, vp, vp, vp, cf, sz, ci, vp]),
    "sqdet_maxpool_nhwc_bwd": (ci, [vp, vp, vp] + [ci] * 8 + [vp]),
    "sqdet_maxpool_nhwc_bwd_relu": (ci, [vp, vp, vp] + [ci] * 8 + [vp]),
    "sqdet_loss_workspace_bytes": (sz, []),
    "sqdet_loss_fwd_bwd": (ci, [vp] * 10 + [ci] * 5 + [cf] * 9 + [ci, vp]),
    "sqdet_loss_fwd_bwd_dev": (ci, [vp] * 10 + [ci] * 5 + [cf] * 8 + [vp, ci, vp]),
    "sqdet_loss_fwd_bwd_mixed": (ci, [vp] * 8 + [cf] + [vp] * 3 + [ci] * 5 + [cf] * 9 + [vp, ci, vp]),
    "sqdet_sum_f32": (ci, [vp, sz, vp, vp]),
    "sqdet_add_relu": (ci, [vp, vp, vp, sz, ci, vp]),
    "sqdet_copy_channels": (ci, [vp, vp, sz, ci, ci, ci, ci, vp]),
    "sqdet_dropout_mask": (ci, [vp, sz, cf, C.c_uint64, ci, vp]),
    "sqdet_optimizer_create": (ci, [C.POINTER(vp), C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(cf), ci]),
    "sqdet_optimizer_destroy": (None, [vp]),
    "sqdet_optimizer_workspace_bytes": (sz, [vp]),
    "sqdet_optimizer_step": (ci, [vp, vp, vp, vp, vp, cf, cf, cf, cf, vp, vp]),
    "sqdet_net_create": (ci, [C.POINTER(vp), ci, ci, ci, ci, ci, ci, ci]),
    "sqdet_net_destroy": (None, [vp]),
    "sqdet_net_num_params": (ci, [vp]),
    "sqdet_net_param_info": (ci, [vp, ci, C.c_char_p, sz, C.POINTER(ci), C.POINTER(ci)]),
    "sqdet_net_param_bytes": (sz, [vp]),
    "sqdet_net_workspace_bytes": (sz, [vp]),
    "sqdet_net_bind": (ci, [vp, vp, vp]),
    "sqdet_net_set_param": (ci, [vp, C.c_char_p, vp, vp]),
    "sqdet_net_set_bn_epsilon": (ci, [vp, cf]),
    "sqdet_net_output_dims": (ci, [vp, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]),
    "sqdet_net_forward": (ci, [vp, vp, vp, vp]),
    "sqdet_net_num_layers": (ci, [vp]),
    "sqdet_net_layer_info": (ci, [vp, ci, C.c_char_p, sz, C.POINTER(cd), C.POINTER(cd)]),
    "sqdet_net_forward_timed": (ci, [vp, vp, vp, C.POINTER(cf), vp]),
    "sqdet_net_set_probe": (ci, [vp, ci, ci]),
    "sqdet_net_read_probe": (ci, [vp, C.POINTER(cf), ci, C.POINTER(ci)]),
    "sqdet_build_labels": (ci, [vp] * 9 + [ci] * 4 + [vp]),
    "sqdet_preprocess_bgr": (ci, [vp, vp] + [ci] * 5 + [cf, cf, cf, ci, vp]),
    "sqdet_copy_to_mapped_host": (ci, [vp, vp, sz, vp]),
    "sqdet_probe_mfma_layout": (ci, [C.POINTER(C.c_int32), ci]),
    "sqdet_calib_mfma": (ci, [vp, sz, ci, C.POINTER(cd), vp]),
    "sqdet_calib_copy": (ci, [vp, vp, sz, vp]),
    "sqdet_calib_mfma2": (ci, [vp, sz, vp, sz, ci, ci, ci, ci, C.POINTER(cd), C.POINTER(ci), vp]),
}


class SqdetError(RuntimeError):
    pass


class SqdetUnsupported(SqdetError):
    """SQDET_EUNSUPPORTED: no kernel of this entry point takes the shape / option combination (the caller may use the
    general entry points instead -- another HIP kernel, never a CPU path)."""


def lib():
    """Loads the library once.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SqdetError(
                "libsqdet_hip.so not found at %s -- build it with `python -m squeezedet_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        # ONE HIP runtime per process: torch ships its own libamdhip64 / libhsa-runtime64, and the device pointers this
        # library is handed come from torch's.  Loaded before torch, libsqdet_hip.so would bind to the system copy under
        # /opt/rocm -- a second runtime that sees no device ("HIP error 100" at the first launch).  With torch imported
        # first the dynamic linker resolves the same sonames to the copy already in the process.
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)  # AttributeError = library/header mismatch: fail loudly
            except AttributeError:
                # (an explicit A/B build -- SQDET_LIB=<older libsqdet_hip_alt.so>, tools/ab_lib.sh -- may predate an entry point: it
                #  stays unbound there and a caller that needs it fails at the call; the in-tree library must export everything)
                if os.environ.get("SQDET_LIB"):
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
        # SQDET_OPTIONS="dbg=200,stem_algo=2": tuning knobs (sqdet_set_option) for A/B runs of unmodified callers (bench.py)
        for kv in filter(None, os.environ.get("SQDET_OPTIONS", "").split(",")):
            k, _, v = kv.partition("=")
            if l.sqdet_set_option(k.strip().encode(), int(v)) != SQDET_OK:
                raise SqdetError("SQDET_OPTIONS: unknown option %r" % kv)
    return _lib


def check(rc, what=""):
    if rc != SQDET_OK:
        msg = lib().sqdet_last_error()
        cls = SqdetUnsupported if rc == SQDET_EUNSUPPORTED else SqdetError
        raise cls("%s failed (code %d): %s" % (what or "sqdet call", rc, msg.decode() if msg else "?"))


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float16:
        return F16
    if torch_dtype == torch.float32:
        return F32
    raise SqdetError("unsupported dtype %s (float16 / float32 only)" % torch_dtype)


def pad_code(padding):
    p = padding.upper()
    if p == "SAME":
        return PAD_SAME
    if p == "VALID":
        return PAD_VALID
    raise SqdetError("padding must be 'SAME' or 'VALID', got %r" % padding)


_raw_stream = None


def stream_ptr():
    """The current torch HIP stream as a hipStream_t value.  (torch.cuda.current_stream() builds a Stream object
    through several Python layers -- 1.6 ms of a 4.7 ms training step's host time; the raw getter is one C call.)"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return C.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
