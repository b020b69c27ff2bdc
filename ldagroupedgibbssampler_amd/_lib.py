"""Loader of libggs_hip.so (the C-ABI declared in include/ggs_hip.h).

There is no CPU fallback: if the shared library is missing or does not load, the
import of anything that needs it raises.  ``build()`` compiles it in-tree with
hipcc for gfx950 (works without a GPU; the .so then travels to the GPU box).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("GGS_HIP_LIB") or os.path.join(CSRC, "libggs_hip.so")   # override: kernel experiments only
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ggs_hip.h")

ABI_VERSION = 5


class GGSConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("num_topics", C.c_int32),
        ("num_types", C.c_int32),
        ("device_id", C.c_int32),
        ("alpha", C.POINTER(C.c_double)),
        ("alpha_scalar", C.c_double),
        ("beta", C.c_double),
        ("seed", C.c_uint64),
        ("flags", C.c_int32),
        ("phi_burn_in", C.c_int32),
        ("phi_mean_thin", C.c_int32),
        ("reserved", C.c_int32),
    ]


class GGSTimings(C.Structure):
    _fields_ = [
        ("theta_ms", C.c_double),
        ("z_ms", C.c_double),
        ("merge_ms", C.c_double),
        ("phi_ms", C.c_double),
        ("sweeps", C.c_int64),
        ("tokens_sampled", C.c_int64),
        ("exchange_ms", C.c_double),
        ("exchange_rs_ms", C.c_double),
        ("exchange_ag_ms", C.c_double),
    ]


# int cb(void *ctx, const void *send, void *recv, int64_t count, void *hip_stream)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


# int cb(void *ctx, const void *send, const int64_t *send_offsets, const int64_t *send_counts, void *recv, const int64_t *recv_offsets,
#        const int64_t *recv_counts, void *hip_stream)
A2AV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                      C.c_void_p)


class GGSExchangeOps(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("reserved", C.c_int32),
        ("ctx", C.c_void_p),
        ("reduce_scatter_i32", EXCHANGE_CB),
        ("all_gather_f64", EXCHANGE_CB),
        ("all_gather_i32", EXCHANGE_CB),
        ("all_to_all_v_i32", A2AV_CB),
    ]


_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
_up = C.POINTER(C.c_uint32)

# name -> (restype, argtypes); must list every symbol include/ggs_hip.h declares
# (tests/test_abi_symbols.py checks the two against each other).
SIGNATURES = {
    "ggs_create": (C.c_int, [C.POINTER(GGSConfig), C.POINTER(_vp)]),
    "ggs_destroy": (None, [_vp]),
    "ggs_last_error": (C.c_char_p, [_vp]),
    "ggs_abi_version": (C.c_int, []),
    "ggs_set_stream": (C.c_int, [_vp, _vp]),
    "ggs_set_corpus": (C.c_int, [_vp, C.c_int64, _lp, _ip, C.c_int64, C.c_int64]),
    "ggs_init_z_java_lcg": (C.c_int, [_vp, C.c_int32]),
    "ggs_java_lcg_next_ints": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _ip]),
    "ggs_set_z": (C.c_int, [_vp, _ip, C.c_int32]),
    "ggs_init_phi": (C.c_int, [_vp]),
    "ggs_set_iteration": (C.c_int, [_vp, C.c_int32]),
    "ggs_get_iteration": (C.c_int, [_vp, _ip]),
    "ggs_sweep": (C.c_int, [_vp, C.c_int32]),
    "ggs_sweep_begin": (C.c_int, [_vp]),
    "ggs_sweep_end": (C.c_int, [_vp]),
    "ggs_sweep_end_async": (C.c_int, [_vp]),
    "ggs_sample_z_given_phi": (C.c_int, [_vp, C.c_int32]),
    "ggs_collapsed_serial_sweep": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "ggs_counts_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), _lp]),
    "ggs_set_global_token_count": (C.c_int, [_vp, C.c_int64]),
    "ggs_synchronize": (C.c_int, [_vp]),
    "ggs_get_z": (C.c_int, [_vp, _ip]),
    "ggs_get_type_topic_counts": (C.c_int, [_vp, _ip]),
    "ggs_get_topic_totals": (C.c_int, [_vp, _ip]),
    "ggs_get_phi": (C.c_int, [_vp, _dp]),
    "ggs_set_phi": (C.c_int, [_vp, _dp]),
    "ggs_get_phi_mean": (C.c_int, [_vp, _dp, _ip]),
    "ggs_get_theta": (C.c_int, [_vp, C.c_int64, C.c_int64, _dp]),
    "ggs_get_doc_topic_counts": (C.c_int, [_vp, C.c_int64, C.c_int64, _ip]),
    "ggs_get_timings": (C.c_int, [_vp, C.POINTER(GGSTimings)]),
    "ggs_reset_timings": (C.c_int, [_vp]),
    "ggs_check_invariants": (C.c_int, [_vp]),
    "ggs_get_launch_info": (C.c_int, [_vp, _lp, _ip, _ip]),
    "ggs_get_num_hot_words": (C.c_int, [_vp, _ip]),
    "ggs_get_z_parts": (C.c_int, [_vp, _ip]),
    "ggs_get_z_form": (C.c_int, [_vp, _ip, _ip, _ip]),
    "ggs_get_warm_tiers": (C.c_int, [_vp, _ip, _ip, _ip]),
    "ggs_attach_exchange": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(GGSExchangeOps)]),
    "ggs_rccl_unique_id": (C.c_int, [_vp]),
    "ggs_attach_rccl": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "ggs_attach_rccl_comm": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "ggs_group_create": (C.c_int, [C.POINTER(GGSConfig), C.c_int32, _ip, C.POINTER(_vp)]),
    "ggs_group_destroy": (None, [C.POINTER(_vp), C.c_int32]),
    "ggs_group_adopt": (C.c_int, [C.POINTER(_vp), C.c_int32]),
    "ggs_group_set_z": (C.c_int, [C.POINTER(_vp), C.c_int32, C.POINTER(_ip), C.c_int32]),
    "ggs_group_sweep": (C.c_int, [C.POINTER(_vp), C.c_int32, C.c_int32]),
    "ggs_group_gather_counts": (C.c_int, [C.POINTER(_vp), C.c_int32]),
    "ggs_attach_null_exchange": (C.c_int, [_vp, C.c_int32, C.c_int32]),
    "ggs_get_exchange_info": (C.c_int, [_vp, _ip, _ip, _ip, _ip]),
    "ggs_get_exchange_provider": (C.c_int, [_vp, _ip, _ip, _ip]),
    "ggs_set_count_exchange": (C.c_int, [_vp, C.c_int32]),
    "ggs_get_count_exchange": (C.c_int, [_vp, _ip, _lp, _lp]),
    "ggs_debug_philox": (C.c_int, [C.c_int32, C.c_int64, _up, _up, _up]),
    "ggs_debug_math": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _dp, _dp, _dp]),
    "ggs_debug_draw": (C.c_int, [C.c_int32, C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int64,
                                 _dp, _dp, _ip]),
    "ggs_model_log_likelihood": (C.c_int, [_vp, _dp, _dp]),
    "ggs_log_posterior": (C.c_int, [_vp, _dp, _dp]),
    "ggs_set_test_corpus": (C.c_int, [_vp, C.c_int64, _lp, _ip, C.c_int64]),
    "ggs_heldout_log_likelihood": (C.c_int, [_vp, C.c_int32, _dp, _dp]),
    "ggs_debug_column_sum": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _ip, C.c_double, _dp]),
    "ggs_debug_column_sum_guided": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _ip, C.c_double, _dp, _dp, _dp, _ip]),
}


def build(force=False):
    """Compile libggs_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))] + [HEADER_PATH]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", CSRC, "libggs_hip.so"])
    return LIB_PATH


_lib = None


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  The torch wheel ships its own
    libamdhip64.so (SONAME libamdhip64.so.7, but libtorch_hip asks for it by the unversioned
    file name), so if libggs_hip.so pulls in /opt/rocm's copy first, a later `import torch`
    loads a second runtime that sees no GPUs.  Loading torch's copy first makes both
    resolve to the same SONAME.  Without torch installed (the JNI deployment) this is a no-op
    and /opt/rocm's runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def share_rccl_with_torch():
    """One process must hold ONE RCCL as well.  libggs_hip dlopen()s "librccl.so.1" on the first use of its multi-GPU
    exchange; if torch is installed its wheel brings a librccl.so of its own (same SONAME), and a process that maps
    /opt/rocm's copy first and torch's later aborts in teardown (double free) -- and so does one that maps torch's copy
    by hand before `import torch`.  The order that works is torch first: then the library's request resolves to the
    copy already mapped.  Without torch (the JNI deployment) this is a no-op and /opt/rocm's RCCL is the only one."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        if importlib.util.find_spec("torch") is None:
            return
    except (ImportError, ValueError):
        return
    import torch  # noqa: F401


def load():
    """dlopen libggs_hip.so and type every entry point.  Raises if it is absent:
    the product has no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libggs_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc); there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = L.ggs_abi_version()
    if got != ABI_VERSION:
        raise ImportError("libggs_hip.so ABI version %d, expected %d" % (got, ABI_VERSION))
    _lib = L
    return L
