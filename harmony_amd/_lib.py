"""ctypes binding of libharmony_mi355x.so (the C ABI in include/harmony_mi355x.h).

There is no CPU implementation behind this module: if the shared library has not been
built (``python -m harmony_amd.build``) or no HIP device is present, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMX_LIB_PATH") or os.path.join(_HERE, "lib", "libharmony_mi355x.so")

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p)
POLL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)

# every symbol include/harmony_mi355x.h declares: name -> (restype, argtypes)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)
SIGNATURES = {
    "hmx_create": (C.c_void_p, []),
    "hmx_destroy": (None, [C.c_void_p]),
    "hmx_last_error": (C.c_char_p, [C.c_void_p]),
    "hmx_last_warning": (C.c_char_p, [C.c_void_p]),
    "hmx_setup": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int32, _ip, _ip, _dp, C.c_int32, _dp, _dp, _dp,
                            C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double,
                            _ip, C.c_int32, C.c_double, C.c_int32]),
    "hmx_setup_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _ip, _ip, _dp, C.c_int32, _dp,
                               _dp, _dp, C.c_int32, C.c_double, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_double,
                               _ip, C.c_int32, C.c_double, C.c_int32]),
    "hmx_restart": (C.c_int, [C.c_void_p]),
    "hmx_init_cluster": (C.c_int, [C.c_void_p, _dp]),
    "hmx_kmeans_centers": (C.c_int, [C.c_void_p, _dp]),
    "hmx_cluster": (C.c_int, [C.c_void_p]),
    "hmx_moe_correct_ridge": (C.c_int, [C.c_void_p]),
    "hmx_check_convergence": (C.c_int, [C.c_void_p, C.c_int32]),
    "hmx_compute_objective": (C.c_int, [C.c_void_p]),
    "hmx_get": (C.c_int64, [C.c_void_p, C.c_char_p, _dp, C.c_int64]),
    "hmx_get_matrix": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64]),
    "hmx_set_int": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "hmx_set_uniform_source": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hmx_r_runif": (None, [C.c_uint32, C.c_int32, _dp]),
    "hmx_r_shuffle": (None, [C.c_uint32, C.c_int64, _lp]),
    "hmx_mt19937_by_array": (None, [C.POINTER(C.c_uint32), C.c_int32, C.c_int32, C.POINTER(C.c_uint32)]),
    "hmx_feistel_pos": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "hmx_feistel_cell": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "hmx_u01": (C.c_float, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "hmx_cluster_of_column": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "hmx_push_update_order": (C.c_int, [C.c_void_p, _lp]),
    "hmx_set_shard": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "hmx_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "hmx_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "hmx_comm_allreduce_host": (C.c_int, [C.c_void_p, _dp, C.c_int32, C.c_int32]),
    "hmx_p2p_export": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8)]),
    "hmx_p2p_connect": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8)]),
    "hmx_p2p_selftest": (C.c_int, [C.c_void_p]),
    "hmx_p2p_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "hmx_p2p_status": (C.c_char_p, [C.c_void_p]),
    "hmx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hmx_set_abort_poll": (C.c_int, [C.c_void_p, POLL_FN, C.c_void_p]),
    "hmx_debug_seq_oe": (C.c_int, [C.POINTER(C.c_float), C.c_int64, C.c_int32, _ip, C.c_int32, _ip, C.c_int64, _ip, _ip, C.c_int32, C.c_int32,
                                   C.c_int32, C.POINTER(C.c_float), _lp, _dp]),
    "hmx_debug_seq_arr": (C.c_int, [C.POINTER(C.c_float), C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), _lp, _dp]),
}

_lib = None


class HarmonyLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HarmonyLibraryError(
            "libharmony_mi355x.so is not built (%s). Run `python -m harmony_amd.build`; "
            "harmony_amd has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
