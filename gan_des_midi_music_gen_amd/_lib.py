"""ctypes binding of libgdm_hip.so (C ABI: include/gdm.h).  No fallback: a missing library is a hard error."""
import ctypes
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
# GDM_LIB_TAG=<tag>: load an experiment build (build.py, GDM_BUILD_TAG) instead of the shipped library; such a build
# reports gdm_build_flavor() == 1 and bench.py refuses it.
_TAG = os.environ.get("GDM_LIB_TAG", "")
LIB_PATH = os.path.join(_PKG, "libgdm_hip" + (f"_{_TAG}" if _TAG else "") + ".so")

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 0, 1, 2, 3

_c = ctypes
_P, _I, _L, _F, _Z = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_size_t

class LinearBnJob(_c.Structure):
    """gdm_linear_bn_job (include/gdm.h)."""
    _fields_ = [(n, _P) for n in ("x", "w", "bias", "gamma", "beta", "running_mean", "running_var",
                                  "num_batches_tracked", "y_out", "out", "save_mean", "save_invstd")] + \
               [(n, _I) for n in ("M", "N", "K", "groups", "stat_repeats")]


class DcnnAdam(_c.Structure):
    """gdm_dcnn_adam (include/gdm.h)."""
    _fields_ = [("param", _P * 6), ("exp_avg", _P * 6), ("exp_avg_sq", _P * 6), ("hyper", _P), ("done", _P)]


class ConcatJob(_c.Structure):
    """gdm_concat_job (include/gdm.h)."""
    _fields_ = [("a", _P), ("b", _P), ("out", _P), ("M", _I), ("Ka", _I), ("Kb", _I)]


# name -> (restype, argtypes); must list every function declared in include/gdm.h (tests/test_abi.py checks that)
SIGNATURES = {
    "gdm_last_error": (_c.c_char_p, []),
    "gdm_version": (_I, []),
    "gdm_arch": (_c.c_char_p, []),
    "gdm_build_flavor": (_I, []),
    "gdm_gemm": (_I, [_P, _I, _L, _L, _P, _I, _L, _L, _P, _I, _L, _L, _I, _I, _I, _P, _P, _I, _F, _I, _I, _P, _Z, _P]),
    "gdm_bce_with_logits": (_I, [_P, _F, _I, _F, _P, _P, _I, _I, _P]),
    "gdm_adam_step": (_I, [_P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P]),
    "gdm_adam_step_dev": (_I, [_P, _P, _P, _P, _L, _P, _P]),
    "gdm_adam_step_dev_pc": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _I, _P]),
    "gdm_stft_frames": (_I, [_P, _I, _L, _L, _I, _I, _I, _P, _P]),
    "gdm_power_spectrum": (_I, [_P, _L, _I, _I, _P, _P]),
    "gdm_power_to_db": (_I, [_P, _I, _I, _I, _F, _F, _P, _P]),
    "gdm_bn_workspace_bytes": (_Z, [_I, _I]),
    "gdm_bn_act_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _F, _F, _I, _P, _I, _P, _P, _I, _P, _Z, _P]),
    "gdm_bn_act_bwd": (_I, [_P, _P, _I, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "gdm_bn_stats": (_I, [_P, _I, _I, _P, _P, _P, _F, _F, _P, _P, _P, _Z, _P]),
    "gdm_bn_partial_chunks": (_I, [_I]),
    "gdm_bn_partials": (_I, [_P, _I, _I, _P, _P]),
    "gdm_bn_apply": (_I, [_P, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P]),
    "gdm_bn_finalize": (_I, [_P, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "gdm_simnn_gen_pack_bytes": (_Z, []),
    "gdm_simnn_gen_pack": (_I, [_P, _I, _P, _P, _P, _P]),
    "gdm_simnn_gen_first": (_I, [_P, _I, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P]),
    "gdm_simnn_gen_convt_chunks": (_I, [_I, _I]),
    "gdm_simnn_gen_convt_bn": (_I, [_I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "gdm_simnn_gen_last": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gdm_bias_act_fwd": (_I, [_P, _P, _I, _I, _I, _F, _P, _I, _P]),
    "gdm_act_bwd": (_I, [_P, _P, _I, _L, _I, _F, _P, _P]),
    "gdm_colsum": (_I, [_P, _I, _I, _I, _P, _P, _Z, _P]),
    "gdm_cast": (_I, [_P, _I, _P, _I, _L, _P]),
    "gdm_nonfinite_count": (_I, [_P, _I, _L, _P, _P]),
    "gdm_simnn_conv1_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "gdm_simnn_conv1_fwd_pair": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "gdm_simnn_conv2_pack_bytes": (_Z, [_I]),
    "gdm_simnn_conv2_pack": (_I, [_P, _I, _P, _P]),
    "gdm_simnn_conv2_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I, _P]),
    "gdm_simnn_conv2_bwd_fused_workspace_bytes": (_Z, [_I, _I, _I]),
    "gdm_simnn_conv2_bwd_fused": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P, _Z, _P]),
    "gdm_simnn_conv2_bwd_fused_finish": (_I, [_I, _I, _I, _P, _P, _P, _Z, _P]),
    "gdm_simnn_conv2_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "gdm_simnn_conv2_bwd_weight_workspace_bytes": (_Z, [_I, _I, _I]),
    "gdm_simnn_conv2_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _Z, _P]),
    "gdm_simnn_conv1_bwd_weight_workspace_bytes": (_Z, [_I, _I, _I]),
    "gdm_simnn_conv1_bwd_weight": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _I, _I, _P, _Z, _P]),
    "gdm_simnn_conv1_bwd_data": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "gdm_simnn_adam_step": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _P]),
    "gdm_simnn_head_workspace_bytes": (_Z, [_I]),
    "gdm_simnn_head": (_I, [_P, _P, _P, _I, _I, _F, _F, _P, _P, _I, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "gdm_linear_bn_act_max_rows": (_I, []),
    "gdm_linear_bn_act_fwd_multi": (_I, [_P, _I, _F, _F, _I, _I, _P]),
    "gdm_concat_cols_multi": (_I, [_P, _I, _P]),
    "gdm_linear_bn_act_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I,
                                   _P]),
    "gdm_dcnn_fused_supported": (_I, [_I]),
    "gdm_dcnn_pack_bytes": (_Z, [_I]),
    "gdm_dcnn_pack": (_I, [_P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gdm_dcnn_fused_workspace_bytes": (_Z, [_I, _I, _I]),
    "gdm_dcnn_fused": (_I, [_P, _I, _P, _P, _I, _I, _F, _F, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "gdm_dcnn_fused_adam": (_I, [_P, _I, _P, _P, _I, _I, _F, _F, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "gdm_im2col": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    "gdm_col2im": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P]),
    "gdm_permute_pc": (_I, [_P, _I, _I, _I, _I, _P, _I, _P]),
    "gdm_des_scan": (_I, [_P, _L, _I, _I, _I, _F, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gdm_des_routing": (_I, [_P, _L, _I, _I, _I, _P, _P, _P, _P]),
    "gdm_des_run": (_I, [_P, _I, _P, _P, _P, _L, _L, _L, _P, _P, _P, _P, _P, _L, _P, _P]),
    "gdm_piano_roll_raster": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gdm_maxpool2_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "gdm_maxpool2_bwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P]),
}

_lock = threading.Lock()
_lib = None


class GdmError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if libgdm_hip.so has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GdmError(
                    f"{LIB_PATH} is missing: build it with `python -m gan_des_midi_music_gen_amd.build` "
                    "(hipcc --offload-arch=gfx950). There is no CPU or eager-PyTorch fallback for this path.")
            # PyTorch-ROCm ships its own libamdhip64: it must be in the process BEFORE this library's dependency on
            # libamdhip64.so is resolved, or the process ends up with two HIP runtimes and our kernels are registered
            # with the one that owns no device ("no ROCm-capable device is detected" on the first launch).
            import torch  # noqa: F401
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the symbol is not exported
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().gdm_last_error().decode(errors="replace")
        raise GdmError(f"{what} failed (rc={rc}): {msg}")
