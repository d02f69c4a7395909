"""ctypes binding of the C ABI in include/ctcdecode_amd.h (the only way the Python layer reaches the decoder).

There is NO CPU fallback: if the HIP library is missing or does not load, importing this module raises.
"""
import ctypes
import os

from . import _build

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)

# ctcd_cond_log10_fn (include/ctcdecode_amd.h): int fn(void *user, const char *const *words, int n, float *log10_prob)
COND_LOG10_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, _f32p)

# ctcd_result_alloc_fn: int fn(void *user, int R, int L, int32_t **tokens, int32_t **timesteps)
RESULT_ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p))

SYMBOLS = ["ctcd_log_softmax", "ctcd_compact_label_capacity", "ctcd_beam_decode_compact", "ctcd_expand_compact", "ctcd_beam_decode_to_host", "ctcd_scorer_create", "ctcd_scorer_destroy", "ctcd_scorer_is_character_based", "ctcd_scorer_max_order", "ctcd_scorer_dict_size",
           "ctcd_scorer_reset_params", "ctcd_scorer_cond_log_prob", "ctcd_scorer_create_callback", "ctcd_scorer_cond_log10", "ctcd_scorer_callback_calls", "ctcd_scorer_callback_seconds", "ctcd_scorer_set_callback_threads", "ctcd_beam_decode_lm", "ctcd_beam_decode_lm_host", "ctcd_stream_create_lm",
           "ctcd_create", "ctcd_destroy", "ctcd_beam_decode", "ctcd_beam_decode_host", "ctcd_check_status", "ctcd_fetch_status_async", "ctcd_stream_create", "ctcd_stream_destroy", "ctcd_stream_frames", "ctcd_stream_decode", "ctcd_stream_decode_to_host", "ctcd_last_prune_host_rows", "ctcd_last_prune_flagged_rows", "ctcd_last_scorer_rounds", "ctcd_last_scorer_waits", "ctcd_set_scorer_wait",
           "ctcd_set_threads", "ctcd_set_cu_sharing", "ctcd_set_subtree_search", "ctcd_last_subtree_search", "ctcd_debug_set_host_path", "ctcd_set_timing", "ctcd_last_kernel_ms", "ctcd_last_prune_ms", "ctcd_debug_math_check", "ctcd_debug_set_profile", "ctcd_debug_set_fixed_layout", "ctcd_debug_set_prune_resolve", "ctcd_debug_set_fused_logits", "ctcd_debug_set_prune_registers", "ctcd_debug_prune_rows", "ctcd_debug_timeline", "ctcd_debug_timeline_cap", "ctcd_debug_get_profile", "ctcd_debug_beam_dump", "ctcd_workgroup_lds_bytes", "ctcd_last_error", "ctcd_version"]


def _load():
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise ImportError("ctcdecode_amd: HIP library %s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(or `python ctcdecode_amd/_build.py`). There is no CPU fallback." % path)
    import torch  # noqa: F401  (loads the HIP runtime the library must share with PyTorch-ROCm)

    lib = ctypes.CDLL(path)
    lib.ctcd_last_error.restype = ctypes.c_char_p
    lib.ctcd_version.restype = ctypes.c_char_p
    lib.ctcd_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
    lib.ctcd_destroy.argtypes = [ctypes.c_void_p]
    lib.ctcd_destroy.restype = None
    lib.ctcd_set_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_set_cu_sharing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_set_subtree_search.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_last_subtree_search.argtypes = [ctypes.c_void_p]
    lib.ctcd_debug_set_host_path.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong]
    lib.ctcd_log_softmax.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ctcd_last_kernel_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.ctcd_last_prune_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    lib.ctcd_debug_math_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
    lib.ctcd_debug_set_profile.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_debug_set_fixed_layout.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_debug_set_prune_resolve.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_debug_set_fused_logits.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_debug_set_prune_registers.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_debug_prune_rows.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ctcd_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ctcd_debug_beam_dump.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.ctcd_debug_get_profile.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_last_scorer_rounds.argtypes = [ctypes.c_void_p]
    lib.ctcd_last_scorer_waits.argtypes = [ctypes.c_void_p]
    lib.ctcd_set_scorer_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_last_scorer_rounds.restype = ctypes.c_int
    lib.ctcd_last_prune_host_rows.argtypes = [ctypes.c_void_p]
    lib.ctcd_last_prune_host_rows.restype = ctypes.c_longlong
    lib.ctcd_last_prune_flagged_rows.argtypes = [ctypes.c_void_p]
    lib.ctcd_last_prune_flagged_rows.restype = ctypes.c_longlong
    lib.ctcd_stream_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ctcd_stream_destroy.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.ctcd_stream_destroy.restype = None
    lib.ctcd_stream_frames.argtypes = [ctypes.c_void_p]
    lib.ctcd_stream_frames.restype = ctypes.c_longlong
    lib.ctcd_stream_decode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.ctcd_stream_decode_to_host.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               RESULT_ALLOC_FN, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    lib.ctcd_check_status.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_fetch_status_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ctcd_workgroup_lds_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    common = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
              ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
              ctypes.c_void_p, ctypes.c_void_p]
    lib.ctcd_beam_decode.argtypes = common + [ctypes.c_void_p]
    lib.ctcd_beam_decode_host.argtypes = common
    lm_common = common[:12] + [ctypes.c_void_p] + common[12:]  # `scorer` sits before the outputs (binding.cpp:122-140)
    lib.ctcd_beam_decode_lm.argtypes = lm_common + [ctypes.c_void_p]
    lib.ctcd_beam_decode_lm_host.argtypes = lm_common
    lib.ctcd_compact_label_capacity.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.ctcd_compact_label_capacity.restype = ctypes.c_longlong
    lib.ctcd_beam_decode_compact.argtypes = common[:12] + [ctypes.c_void_p] * 5 + [ctypes.c_longlong] + [ctypes.c_void_p] * 4
    lib.ctcd_expand_compact.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    lib.ctcd_beam_decode_to_host.argtypes = common[:3] + [ctypes.c_int] + common[3:12] + [ctypes.c_void_p] + common[12:] + [ctypes.c_void_p]
    lib.ctcd_scorer_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_double, ctypes.c_double, ctypes.c_char_p,
                                       ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int]
    lib.ctcd_scorer_destroy.argtypes = [ctypes.c_void_p]
    lib.ctcd_scorer_destroy.restype = None
    for name in ("ctcd_scorer_is_character_based", "ctcd_scorer_max_order", "ctcd_scorer_dict_size"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.ctcd_scorer_reset_params.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
    lib.ctcd_scorer_cond_log_prob.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    lib.ctcd_scorer_cond_log_prob.restype = ctypes.c_double
    lib.ctcd_scorer_create_callback.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_char_p),
                                                ctypes.c_int, COND_LOG10_FN, ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, ctypes.c_int]
    lib.ctcd_scorer_cond_log10.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int, _f32p]
    lib.ctcd_scorer_callback_calls.argtypes = [ctypes.c_void_p]
    lib.ctcd_scorer_callback_calls.restype = ctypes.c_longlong
    lib.ctcd_scorer_callback_seconds.argtypes = [ctypes.c_void_p]
    lib.ctcd_scorer_callback_seconds.restype = ctypes.c_double
    lib.ctcd_scorer_set_callback_threads.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.ctcd_scorer_set_callback_threads.restype = ctypes.c_int
    lib.ctcd_stream_create_lm.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


lib = _load()


class NativeError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib.ctcd_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError("ctcdecode_amd: " + msg)
        if rc == -2:
            raise NotImplementedError("ctcdecode_amd: " + msg)
        raise NativeError("ctcdecode_amd (code %d): %s" % (rc, msg))
