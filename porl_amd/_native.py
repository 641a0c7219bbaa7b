"""ctypes binding of libporl_hip.so (C ABI in include/porl_hip.h).

There is NO CPU fallback: if the library is missing or does not load, importing a compute entry
point raises.  Build it with `python -m porl_amd.build` (or `__graft_entry__.build()`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libporl_hip.so")

ABI_VERSION = 5

# every symbol include/porl_hip.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "porl_abi_version", "porl_last_error",
    "porl_iql_create", "porl_iql_destroy", "porl_iql_group_floats", "porl_iql_group_tensors",
    "porl_iql_tensor_info", "porl_iql_workspace_floats", "porl_iql_bind", "porl_iql_load_batch",
    "porl_iql_load_batch_sampled", "porl_iql_set_stats", "porl_iql_set_mode",
    "porl_iql_value_backward", "porl_iql_value_apply", "porl_iql_policy_forward", "porl_iql_policy_backward",
    "porl_iql_policy_apply", "porl_iql_step", "porl_iql_policy_prefetch", "porl_iql_forward_value", "porl_iql_forward_policy",
    "porl_gemm_f32", "porl_adam_ema", "porl_ema", "porl_softmax_mask", "porl_gather_rows", "porl_sample_indices", "porl_epoch_indices", "porl_per_update", "porl_per_sample",
    "porl_prof_enable", "porl_prof_read", "porl_tune_set", "porl_tune_set_ptr", "porl_state2costmap",
    "porl_signal_create", "porl_signal_destroy", "porl_signal_write", "porl_signal_wait_ge", "porl_iql_update_pipelined",
    "porl_qnet_create", "porl_qnet_destroy", "porl_qnet_param_floats", "porl_qnet_tensors",
    "porl_qnet_tensor_info", "porl_qnet_workspace_floats", "porl_qnet_bind", "porl_qnet_load_batch",
    "porl_qnet_cql_backward", "porl_qnet_apply", "porl_qnet_learn", "porl_qnet_sync_target",
    "porl_qnet_forward", "porl_qnet_forward_loaded", "porl_qnet_backward", "porl_qr_loss", "porl_iqn_quantile_huber", "porl_iqn_cos_embed", "porl_iqn_hadamard", "porl_iqn_hadamard_backward", "porl_iqn_select", "porl_iqn_scatter", "porl_iqn_target", "porl_grad_clip", "porl_c51_loss", "porl_reduce_mean", "porl_qnet_penalty", "porl_qnet_learn_indexed", "porl_qnet_one_launch", "porl_qnet_learn_variant", "porl_qnet_can_sample", "porl_qnet_learn_sampled",
    "porl_enc_create", "porl_enc_destroy", "porl_enc_param_floats", "porl_enc_stat_floats",
    "porl_enc_workspace_floats", "porl_enc_tensors", "porl_enc_norms", "porl_enc_blocks",
    "porl_enc_tensor_info", "porl_enc_norm_info", "porl_enc_bind", "porl_enc_weights_changed", "porl_enc_forward",
]


class IqlCfg(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("pol_out_dim", C.c_int32), ("hidden_dim", C.c_int32),
                ("n_hidden", C.c_int32), ("layer_norm", C.c_int32), ("pol_tanh", C.c_int32),
                ("weight_mode", C.c_int32), ("max_batch", C.c_int32)]


class IqlBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("params_vf", "params_tgt", "params_pol", "grads_vf", "grads_pol", "adam_m_vf",
                 "adam_v_vf", "adam_m_pol", "adam_v_pol", "workspace", "stats")]


class IqlHyper(C.Structure):
    _fields_ = [("tau", C.c_float), ("discount", C.c_float), ("alpha", C.c_float), ("inv_batch", C.c_float),
                ("value_step", C.c_int32), ("policy_step", C.c_int32), ("reserved", C.c_int32),
                ("ema_beta", C.c_double), ("value_lr", C.c_double), ("policy_lr", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double)]


class QnetCfg(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("n_actions", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * 8), ("max_batch", C.c_int32)]


class QnetBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("params", "params_tgt", "grads", "adam_m", "adam_v", "workspace", "stats")]


class QnetHyper(C.Structure):
    _fields_ = [("gamma", C.c_float), ("alpha", C.c_float), ("inv_batch", C.c_float), ("step", C.c_int32),
                ("lr", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double)]


class QnetVariant(C.Structure):
    _fields_ = [("double_dqn", C.c_int32), ("is_weights", C.c_void_p), ("uniform_weight", C.c_void_p),
                ("td_abs", C.c_void_p), ("next_mask", C.c_void_p), ("td_off", C.c_int32)]


class EncCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_ang", "n_dist", "embed_dim", "depth0", "depth1", "n_div",
                                         "feature_dim", "num_classes", "max_batch")] + \
               [("mlp_ratio", C.c_float), ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("bf16_operands", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


class NativeError(RuntimeError):
    pass


_lib = None


def _declare(lib):
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
    lib.porl_abi_version.restype = C.c_int
    lib.porl_last_error.restype = C.c_char_p
    lib.porl_iql_create.argtypes = [C.POINTER(IqlCfg), C.POINTER(vp)]
    lib.porl_iql_destroy.argtypes = [vp]
    lib.porl_iql_destroy.restype = None
    lib.porl_iql_group_floats.argtypes = [vp, C.c_int]
    lib.porl_iql_group_floats.restype = i64
    lib.porl_iql_group_tensors.argtypes = [vp, C.c_int]
    lib.porl_iql_group_tensors.restype = i32
    lib.porl_iql_tensor_info.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]
    lib.porl_iql_workspace_floats.argtypes = [vp]
    lib.porl_iql_workspace_floats.restype = i64
    lib.porl_iql_bind.argtypes = [vp, C.POINTER(IqlBuffers)]
    lib.porl_iql_load_batch.argtypes = [vp, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    lib.porl_iql_load_batch_sampled.argtypes = [vp, i32, vp, i64, i64, i32, i32, C.c_uint64, C.c_uint64, vp, vp]
    lib.porl_iql_set_stats.argtypes = [vp, vp]
    lib.porl_iql_set_mode.argtypes = [vp, i32]
    for name in ("porl_iql_value_backward", "porl_iql_value_apply", "porl_iql_policy_forward", "porl_iql_policy_backward",
                 "porl_iql_policy_apply", "porl_iql_step"):
        getattr(lib, name).argtypes = [vp, C.POINTER(IqlHyper), vp]
    lib.porl_iql_policy_prefetch.argtypes = [vp, vp]
    lib.porl_iql_forward_value.argtypes = [vp, C.c_int, vp, i64, i32, vp, vp, vp]
    lib.porl_iql_forward_policy.argtypes = [vp, vp, i64, i32, vp, i64, vp]
    lib.porl_gemm_f32.argtypes = [C.c_int, C.c_int, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, C.c_int,
                                  vp, i32, C.c_int, vp, vp]
    lib.porl_adam_ema.argtypes = [vp, vp, vp, vp, vp, i64, f64, i32, f64, f64, f64, f64, vp]
    lib.porl_ema.argtypes = [vp, vp, i64, f64, vp]
    lib.porl_softmax_mask.argtypes = [vp, i64, i32, i32, f32, i32, vp, vp]
    lib.porl_gather_rows.argtypes = [vp, i64, vp, i32, i32, vp, i64, vp]
    lib.porl_sample_indices.argtypes = [i64, i32, C.c_uint64, C.c_uint64, i64, vp, vp]
    lib.porl_epoch_indices.argtypes = [i64, i64, i32, C.c_uint64, C.c_uint64, i64, vp, vp]
    lib.porl_per_update.argtypes = [vp, i64, vp, vp, i32, f64, f64, vp, vp]
    lib.porl_per_sample.argtypes = [vp, i64, vp, i32, i64, f64, vp, vp, vp, vp]
    lib.porl_tune_set.argtypes = [C.c_char_p, C.c_int]
    lib.porl_tune_set_ptr.argtypes = [C.c_char_p, vp]
    lib.porl_signal_create.argtypes = [C.POINTER(vp)]
    lib.porl_signal_destroy.argtypes = [vp]
    lib.porl_signal_write.argtypes = [vp, C.c_uint64, vp]
    lib.porl_signal_wait_ge.argtypes = [vp, C.c_uint64, vp]
    lib.porl_iql_update_pipelined.argtypes = [vp, C.POINTER(IqlHyper), i32, vp, i64, i64, i32, i32, C.c_uint64, C.c_uint64,
                                              vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, i32, vp, vp]
    lib.porl_state2costmap.argtypes = [vp, i64, i32, i32, i32, vp, vp]
    lib.porl_qnet_create.argtypes = [C.POINTER(QnetCfg), C.POINTER(vp)]
    lib.porl_qnet_destroy.argtypes = [vp]
    lib.porl_qnet_destroy.restype = None
    lib.porl_qnet_param_floats.argtypes = [vp]
    lib.porl_qnet_param_floats.restype = i64
    lib.porl_qnet_tensors.argtypes = [vp]
    lib.porl_qnet_tensors.restype = i32
    lib.porl_qnet_tensor_info.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.porl_qnet_workspace_floats.argtypes = [vp]
    lib.porl_qnet_workspace_floats.restype = i64
    lib.porl_qnet_bind.argtypes = [vp, C.POINTER(QnetBuffers)]
    lib.porl_qnet_load_batch.argtypes = [vp, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp]
    for name in ("porl_qnet_cql_backward", "porl_qnet_apply", "porl_qnet_learn"):
        getattr(lib, name).argtypes = [vp, C.POINTER(QnetHyper), vp]
    lib.porl_qnet_learn_indexed.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp, i32, C.POINTER(QnetHyper), vp]
    lib.porl_qnet_learn_variant.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, vp, i32, C.POINTER(QnetHyper),
                                            C.POINTER(QnetVariant), vp]
    lib.porl_qnet_one_launch.argtypes = [vp]
    lib.porl_qnet_one_launch.restype = i32
    lib.porl_qnet_can_sample.argtypes = [vp]
    lib.porl_qnet_can_sample.restype = i32
    lib.porl_qnet_learn_sampled.argtypes = [vp, vp, i64, vp, vp, vp, i64, vp, i64, C.c_uint64, C.c_uint64, i32,
                                            C.POINTER(QnetHyper), vp]
    lib.porl_qnet_sync_target.argtypes = [vp, vp]
    lib.porl_qnet_forward.argtypes = [vp, C.c_int, vp, i64, i32, vp, i64, vp]
    lib.porl_qnet_forward_loaded.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, i64, vp]
    lib.porl_qnet_backward.argtypes = [vp, vp, i64, vp]
    lib.porl_qr_loss.argtypes = [vp, vp, vp, i64, vp, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp]
    lib.porl_c51_loss.argtypes = [vp, vp, i64, vp, vp, vp, vp, i32, i32, i32, f32, f32, f32, vp, vp, vp]
    lib.porl_reduce_mean.argtypes = [vp, i32, vp, vp]
    lib.porl_iqn_quantile_huber.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, vp, vp]
    lib.porl_iqn_cos_embed.argtypes = [vp, i64, i32, vp, vp]
    lib.porl_iqn_hadamard.argtypes = [vp, i64, vp, i32, i32, i32, vp, vp]
    lib.porl_iqn_hadamard_backward.argtypes = [vp, vp, i64, vp, i32, i32, i32, vp, vp, vp]
    lib.porl_iqn_select.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.porl_iqn_scatter.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.porl_iqn_target.argtypes = [vp, vp, vp, vp, f32, i32, i32, i32, vp, vp, vp]
    lib.porl_grad_clip.argtypes = [vp, i64, f32, vp, vp, vp]
    lib.porl_qnet_penalty.argtypes = [vp, vp, i64, vp, i64, i32, vp, vp]
    lib.porl_enc_create.argtypes = [C.POINTER(EncCfg), C.POINTER(vp)]
    lib.porl_enc_destroy.argtypes = [vp]
    lib.porl_enc_destroy.restype = None
    for name in ("porl_enc_param_floats", "porl_enc_stat_floats", "porl_enc_workspace_floats"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = i64
    for name in ("porl_enc_tensors", "porl_enc_norms", "porl_enc_blocks"):
        getattr(lib, name).argtypes = [vp]
        getattr(lib, name).restype = i32
    lib.porl_enc_tensor_info.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), C.c_char_p, i32]
    lib.porl_enc_norm_info.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.c_char_p, i32]
    lib.porl_enc_bind.argtypes = [vp, vp, vp, vp]
    lib.porl_enc_weights_changed.argtypes = [vp]
    lib.porl_enc_forward.argtypes = [vp, vp, i64, i32, i32, vp, vp, i64, vp]
    lib.porl_prof_enable.argtypes = [C.c_int]
    lib.porl_prof_read.argtypes = [C.POINTER(ProfEntry), C.c_int]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("porl_abi_version",):
            fn.restype = C.c_int


def lib():
    """Load (once) and return the shared library; raises NativeError when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} not found — run `python -m porl_amd.build` "
                              "(hipcc --offload-arch=gfx950); porl_amd has no CPU fallback")
        # PyTorch must load ITS HIP runtime first: the wheel bundles libamdhip64.so.7 / libhsa-runtime64 of its own ROCm
        # build, the library is linked against /opt/rocm's copies with the same SONAMEs, and a process that ends up with
        # both sets (our library first, torch second) launches into a runtime that sees no device
        # ("no ROCm-capable device is detected").  Loaded in this order, both share torch's copy.
        import torch  # noqa: F401
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        _declare(l)
        v = l.porl_abi_version()
        if v != ABI_VERSION:
            raise NativeError(f"libporl_hip.so ABI {v} != expected {ABI_VERSION}; rebuild")
        _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().porl_last_error().decode("utf-8", "replace")
        raise NativeError(f"{what or 'porl_hip'} failed (rc={rc}): {msg}")


def ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr(device=None):
    """torch's current stream ON `device` (a torch.device / tensor / None = the thread's current device).  The C
    entry points switch to the device that owns their buffers themselves; the stream must belong to that device."""
    import torch
    if device is not None and not isinstance(device, torch.device):
        device = device.device
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
