"""ctypes binding of libstylerenderer_hip.so (the C ABI declared in include/stylerenderer_amd.h).

The product path for device tensors goes through this module only.  There is no fallback:
if the shared library is missing or does not export a declared symbol, `lib()` raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STYLERENDERER_AMD_LIB: development override (ablation builds of scripts/build_variant.sh)
LIB_PATH = os.environ.get("STYLERENDERER_AMD_LIB") or os.path.join(_HERE, "libstylerenderer_hip.so")

_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float
_d = ctypes.c_double
_p = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/stylerenderer_amd.h line by line
SIGNATURES = {
    "sr_error_string": (ctypes.c_char_p, [_i]),
    "sr_abi_version": (_i, []),
    "sr_fused_bias_act": (_i, [_p, _p, _p, _p, _i, _i, _f, _f, _l, _l, _l, _i, _i, _p]),
    "sr_fused_act_bwd_scratch_floats": (_l, [_l, _l, _l]),
    "sr_fused_act_bwd": (_i, [_p, _p, _p, _p, _f, _f, _l, _l, _l, _p, _p]),
    "sr_noise_bias_act": (_i, [_p] * 6 + [_f, _f] + [_l] * 4 + [_p]),
    "sr_noise_bias_act_bwd_scratch_floats": (_l, [_l, _l, _l]),
    "sr_noise_bias_act_bwd": (_i, [_p] * 6 + [_f, _f] + [_l] * 4 + [_p, _p]),
    "sr_noise_bias_act_bwd_dot_scratch_floats": (_l, [_l, _l, _l]),
    "sr_noise_bias_act_bwd_dot": (_i, [_p] * 9 + [_f, _f] + [_l] * 4 + [_p, _p]),
    "sr_conv2d_nba": (_i, [_p] * 8 + [_f, _f] + [_l] * 7 + [_p, _p]),
    "sr_conv2d_nba_ex": (_i, [_p] * 8 + [_f, _f] + [_l] * 7 + [_i, _p, _p]),
    "sr_noise_bias_act_affine": (_i, [_p] * 4 + [_l] + [_p] * 3 + [_f, _f] + [_l] * 4 + [_p]),
    "sr_noise_bias_act_affine_bwd_scratch_floats": (_l, [_l, _l, _l]),
    "sr_noise_bias_act_affine_bwd": (_i, [_p] * 9 + [_l, _p, _f, _f] + [_l] * 4 + [_p, _p]),
    "sr_noise_bias_act_affine_bwd2_scratch_floats": (_l, [_l, _l, _l]),
    "sr_noise_bias_act_affine_bwd2": (_i, [_p, _p, _p, _l, _p, _p, _l] + [_p] * 6 + [_l, _p, _f, _f] + [_l] * 4 + [_p, _p]),
    "sr_adam_flat": (_i, [_p] * 4 + [_l] + [_f] * 4 + [_p, _p]),
    "sr_adam_flat_guarded": (_i, [_p] * 4 + [_l] + [_f] * 4 + [_p, ctypes.POINTER(_l), _i, _p, _p]),
    "sr_rowdot_scratch_floats": (_l, [_l, _l]),
    "sr_rowdot": (_i, [_p] * 5 + [_l, _l, _p, _p]),
    "sr_rowdot_div": (_i, [_p] * 4 + [_l, _l, _p, _p]),
    "sr_rowdot_bwd": (_i, [_p] * 8 + [_l, _l, _p, _p]),
    "sr_smallconv_fwd": (_i, [_p] * 4 + [_l] * 4 + [_p]),
    "sr_smallconv_dx": (_i, [_p] * 3 + [_l] * 4 + [_p]),
    "sr_smallconv_dx_add": (_i, [_p] * 4 + [_l] * 4 + [_p]),
    "sr_smallconv_dw_scratch_floats": (_l, [_l] * 4),
    "sr_smallconv_dw": (_i, [_p] * 3 + [_l] * 4 + [_p, _p]),
    "sr_smallconv_dw_bias": (_i, [_p] * 4 + [_l] * 4 + [_p, _p]),
    "sr_modrows_fwd": (_i, [_p] * 3 + [_f] + [_l] * 3 + [_p]),
    "sr_modrows_bwd": (_i, [_p] * 5 + [_f] + [_l] * 3 + [_p]),
    "sr_vertex_normals_f32": (_i, [_p] * 6 + [_l] * 3 + [_f, _p]),
    "sr_linear_fwd": (_i, [_p] * 4 + [_l] * 4 + [_f, _f, _i, _f, _f, _p]),
    "sr_linear_bwd_x": (_i, [_p] * 4 + [_l] * 3 + [_f, _i, _f, _f, _p]),
    "sr_linear_bwd_w": (_i, [_p] * 5 + [_l] * 4 + [_f, _f, _i, _f, _f, _p]),
    "sr_demod_fwd": (_i, [_p] * 3 + [_l] * 3 + [_f, _p]),
    "sr_demod_bwd": (_i, [_p] * 7 + [_l] * 3 + [_p]),
    "sr_weight_prep": (_i, [_p, _p, _p, _f, _l, _l, _i, _l, _p]),
    "sr_weight_prep_bwd": (_i, [_p, _p, _p, _p, _f, _l, _l, _i, _l, _p]),
    "sr_weight_adjoint": (_i, [_p, _p, _l, _l, _l, _l, _l, _i, _p]),
    "sr_bank_nt": (_i, [_i] + [_p] * 8 + [_l, _f, _f, _p]),
    "sr_bank_nn": (_i, [_i] + [_p] * 8 + [_l, _f, _p]),
    "sr_bank_tn": (_i, [_i] + [_p] * 8 + [_l, _f, _f, _p]),
    "sr_weight_prep_batch": (_i, [_i] + [_p] * 9),
    "sr_weight_adjoint_batch": (_i, [_i] + [_p] * 9),
    "sr_lpips_layer_scratch_floats": (_l, [_l, _l]),
    "sr_lpips_layer_fwd": (_i, [_p] * 4 + [_l] * 4 + [_f, _p, _p]),
    "sr_lpips_layer_bwd": (_i, [_p] * 5 + [_l] * 4 + [_f, _p]),
    "sr_mse_fwd": (_i, [_p] * 3 + [_l, _p]),
    "sr_mse_bwd": (_i, [_p] * 4 + [_l, _p]),
    "sr_maxpool2_fwd": (_i, [_p] * 2 + [_l] * 3 + [_p]),
    "sr_maxpool2_bwd": (_i, [_p] * 3 + [_l] * 3 + [_p]),
    "sr_upfirdn2d": (_i, [_p, _p, _p, _l] + [_i] * 14 + [_p]),
    "sr_upsample2_add": (_i, [_p] * 4 + [_l] + [_i] * 6 + [_p]),
    "sr_blur_noise_bias_act": (_i, [_p] * 6 + [_f, _f, _l, _l] + [_i] * 6 + [_l, _p]),
    "sr_conv2d_generic": (_i, [_p] * 4 + [_l] * 7 + [_i] * 6 + [_p]),
    "sr_conv2d_generic_dgrad": (_i, [_p] * 3 + [_l] * 7 + [_i] * 6 + [_p]),
    "sr_conv2d_generic_wgrad": (_i, [_p] * 3 + [_l] * 7 + [_i] * 6 + [_p]),
    "sr_blur_nba_bwd_scratch_floats": (_l, [_l, _l, _i, _i]),
    "sr_blur_nba_bwd": (_i, [_p] * 10 + [_f, _f, _l, _l] + [_i] * 5 + [_l, _p, _p]),
    "sr_rasterize_scratch_bytes": (_l, [_l, _l, _l, _l, _i]),
    "sr_rasterize_forward_f32": (_i, [_l] * 5 + [_i] * 3 + [_p] * 5 + [_f, _p, _l, _p, _p, _p, _p, _p]),
    "sr_rasterize_levels_supported": (_i, [_i, _l, _l, _p, _p]),
    "sr_rasterize_forward_levels_f32": (_i, [_i, _l, _l, _l, _p, _p, _i, _i, _i, _p, _p, _f, _p, _l, _p, _p, _p, _p, _p]),
    "sr_rasterize_grad_levels_f32": (_i, [_i, _l, _l, _l, _p, _p, _i, _i, _p, _p, _l] + [_p] * 6 + [_l, _l] + [_p] * 3 + [_f, _p, _p]),
    "sr_rasterize_forward_f64": (_i, [_l] * 5 + [_i] * 3 + [_p] * 5 + [_d, _p, _l, _p, _p, _p, _p, _p]),
    "sr_rasterize_grad_scratch_bytes": (_l, [_l, _l, _l, _i]),
    "sr_rasterize_forward_cpu_f32": (_i, [_l] * 5 + [_i] * 3 + [_p] * 5 + [_f]),
    "sr_rasterize_forward_cpu_f64": (_i, [_l] * 5 + [_i] * 3 + [_p] * 5 + [_d]),
    "sr_rasterize_backward_cpu_f32": (_i, [_l] * 4 + [_i] + [_p] * 3 + [_f]),
    "sr_rasterize_backward_cpu_f64": (_i, [_l] * 4 + [_i] + [_p] * 3 + [_d]),
    "sr_rasterize_backward_f32": (_i, [_l] * 4 + [_i] * 2 + [_p] * 3 + [_f, _p]),
    "sr_rasterize_backward_f64": (_i, [_l] * 4 + [_i] * 2 + [_p] * 3 + [_d, _p]),
    "sr_rasterize_grad_f32": (_i, [_l] * 5 + [_i] * 2 + [_p, _p, _l] + [_p] * 6 + [_l, _l, _p, _p, _p, _f, _p, _p]),
    "sr_conv2d_wgrad_scratch_floats": (_l, [_l] * 7 + [_i] * 4),
    "sr_conv2d_wgrad_mfma": (_i, [_p] * 5 + [_l] * 7 + [_i] * 4 + [_p, _p]),
    "sr_conv2d_scratch_floats": (_l, [_l] * 7 + [_i] * 4),
    "sr_conv2d_mfma": (_i, [_p] * 6 + [_l] * 8 + [_i] * 4 + [_p, _p]),
    "sr_conv2d_uses_winograd": (_i, [_l] * 5 + [_p, _p]),
    "sr_conv2d_mfma_ex": (_i, [_p] * 6 + [_l] * 8 + [_i] * 5 + [_p, _p]),
    "sr_conv1x1_add_supported": (_i, [_l] * 5 + [_p] * 4),
    "sr_conv1x1_add": (_i, [_p] * 5 + [_l] * 5 + [_p]),
    "sr_rasterize_grad_f64": (_i, [_l] * 5 + [_i] * 2 + [_p, _p, _l] + [_p] * 6 + [_l, _l, _p, _p, _p, _d, _p, _p]),
    "sr_pose_fwd": (_i, [_p, _p, _p, _p]),
    "sr_pose_bwd": (_i, [_p, _p, _p, _p, _p]),
    "sr_pose_batch_fwd": (_i, [_p, _p, _p, _l, _p]),
    "sr_affine3_fwd": (_i, [_p, _p, _p, _p, _l, _l, _l, _p]),
    "sr_affine3_bwd": (_i, [_p, _p, _p, _p, _l, _l, _l, _p]),
    "sr_signal_bump": (_i, [_p, _p]),
    "sr_signal_wait": (_i, [_p, ctypes.c_uint32, _p]),
    "sr_signal_set": (_i, [_p, _p, _p]),
    "sr_signal_set_host": (_i, [_p, _p, _p]),
    "sr_signal_wait_timeout": (_i, [_p, ctypes.c_uint32, ctypes.c_uint64, _p, _i, _p]),
    "sr_signal_wait_poison": (_i, [_p, ctypes.c_uint32, ctypes.c_uint64, _p, _i, _p, _p]),
    "sr_graph_replace_memset_nodes": (_i, [_p, ctypes.POINTER(_i)]),
    "sr_graph_node_count": (_i, [_p, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Loads the HIP library once.  Raises NativeLibraryError when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(
            "stylerenderer_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for device "
            "tensors." % LIB_PATH)
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError("stylerenderer_amd: cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError:
            raise NativeLibraryError("stylerenderer_amd: %s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().sr_error_string(rc)
        raise RuntimeError("%s failed: %s (code %d)" % (what or "stylerenderer_amd call",
                                                        msg.decode() if msg else "?", rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Zero-size tensors map to NULL."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def current_stream(device=None):
    import torch

    return torch.cuda.current_stream(device).cuda_stream
