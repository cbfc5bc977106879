"""ctypes binding of libmiosqp_hip.so (the C ABI declared in include/miosqp_amd.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded this
module raises, and `OSQP.setup` raises when no gfx950 device is visible.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmiosqp_hip.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class Settings(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf")] + \
               [(k, C.c_int32) for k in ("max_iter", "scaling", "check_termination", "warm_start",
                                         "device", "max_batch", "fold", "resident", "setup_on_device", "coop", "pers", "batch_pers",
                                         "rho_auto")]


class Info(C.Structure):
    _fields_ = [("status_val", C.c_int32), ("iter", C.c_int32), ("run_time", C.c_double),
                ("obj_val", C.c_double), ("pri_res", C.c_double), ("dua_res", C.c_double),
                ("device_time", C.c_double), ("lower", C.c_double),
                ("int_inf", C.c_int32), ("nextvar", C.c_int32), ("heur_viol", C.c_double),
                ("heur_obj", C.c_double)]


class TreeInfo(C.Structure):
    _fields_ = [("nodes", C.c_int32), ("osqp_iter", C.c_int32), ("leaves_left", C.c_int32), ("overflow", C.c_int32),
                ("max_leaves", C.c_int32), ("found", C.c_int32), ("upper_glob", C.c_double), ("lower_glob", C.c_double),
                ("device_time", C.c_double), ("run_time", C.c_double)]


class SearchInfo(C.Structure):
    _fields_ = [("nodes", C.c_int64), ("osqp_iter", C.c_int64), ("open_leaves", C.c_int32), ("free_slots", C.c_int32),
                ("improved", C.c_int32), ("reserved", C.c_int32), ("upper_glob", C.c_double), ("lower_glob", C.c_double),
                ("device_time", C.c_double), ("run_time", C.c_double)]


class StreamInfo(C.Structure):
    _fields_ = [("alive", C.c_int64), ("open_leaves", C.c_int64), ("in_flight", C.c_int64), ("free_slots", C.c_int64),
                ("nodes", C.c_int64), ("osqp_iter", C.c_int64), ("chunks", C.c_int64), ("dropped", C.c_int64),
                ("improved", C.c_int32), ("active", C.c_int32), ("upper_glob", C.c_double)]


class PoolDigest(C.Structure):
    _fields_ = [("slot", C.c_int32), ("status_val", C.c_int32), ("iter", C.c_int32), ("int_inf", C.c_int32),
                ("nextvar", C.c_int32), ("reserved", C.c_int32), ("lower", C.c_double), ("heur_viol", C.c_double),
                ("heur_obj", C.c_double), ("pri_res", C.c_double), ("dua_res", C.c_double)]


# every symbol include/miosqp_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "miosqp_qp_default_settings": (C.c_int, [C.POINTER(Settings)]),
    "miosqp_qp_constant": (C.c_int, [C.c_char_p]),
    "miosqp_qp_setup": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, ip, ip, dp, ip, ip,
                                  dp, dp, dp, dp, C.POINTER(Settings)]),
    "miosqp_qp_update_bounds": (C.c_int, [C.c_void_p, dp, dp]),
    "miosqp_qp_update_lin_cost": (C.c_int, [C.c_void_p, dp]),
    "miosqp_qp_warm_start": (C.c_int, [C.c_void_p, dp, dp]),
    "miosqp_qp_solve": (C.c_int, [C.c_void_p, dp, dp, C.POINTER(Info)]),
    "miosqp_qp_set_integer_rows": (C.c_int, [C.c_void_p, C.c_int32, ip, C.c_int32]),
    "miosqp_qp_set_root": (C.c_int, [C.c_void_p, dp, dp, C.c_double, C.c_double]),
    "miosqp_qp_solve_node": (C.c_int, [C.c_void_p, dp, dp, dp, dp, dp, dp, C.POINTER(Info)]),
    "miosqp_qp_solve_trees": (C.c_int, [C.c_void_p, C.c_int32, dp, dp, dp, dp, dp, dp, dp, C.c_int32, C.c_int32, dp,
                                        C.POINTER(TreeInfo)]),
    "miosqp_qp_search_create": (C.c_int, [C.c_void_p, C.c_int32]),
    "miosqp_qp_search_reset": (C.c_int, [C.c_void_p]),
    "miosqp_qp_search_add_leaf": (C.c_int, [C.c_void_p, dp, dp, dp, dp, C.c_int32, C.c_double]),
    "miosqp_qp_search_take_leaf": (C.c_int, [C.c_void_p, dp, dp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "miosqp_qp_search_set_incumbent": (C.c_int, [C.c_void_p, C.c_double, dp]),
    "miosqp_qp_search_get_incumbent": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), dp]),
    "miosqp_qp_search_run": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.POINTER(SearchInfo)]),
    "miosqp_qp_stream_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "miosqp_qp_stream_begin": (C.c_int, [C.c_void_p]),
    "miosqp_qp_stream_add_leaf": (C.c_int, [C.c_void_p, dp, dp, dp, dp, C.c_int32, C.c_double]),
    "miosqp_qp_stream_take_leaf": (C.c_int, [C.c_void_p, dp, dp, dp, dp, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "miosqp_qp_stream_set_incumbent": (C.c_int, [C.c_void_p, C.c_double, dp]),
    "miosqp_qp_stream_get_incumbent": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), dp]),
    "miosqp_qp_stream_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.POINTER(StreamInfo)]),
    "miosqp_qp_solve_batch": (C.c_int, [C.c_void_p, C.c_int32, dp, dp, dp, dp, dp, dp,
                                        C.POINTER(Info)]),
    "miosqp_qp_solve_tree": (C.c_int, [C.c_void_p, dp, dp, dp, dp, C.c_double, dp, C.c_int32, C.c_int32, dp,
                                       C.POINTER(TreeInfo)]),
    "miosqp_qp_pool_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "miosqp_qp_pool_reset": (C.c_int, [C.c_void_p]),
    "miosqp_qp_pool_write_node": (C.c_int, [C.c_void_p, C.c_int32, dp, dp, dp, dp]),
    "miosqp_qp_pool_read_node": (C.c_int, [C.c_void_p, C.c_int32, dp, dp, dp, dp]),
    "miosqp_qp_pool_push": (C.c_int, [C.c_void_p, C.c_int32, ip, ip, ip, dp]),
    "miosqp_qp_pool_set_upper": (C.c_int, [C.c_void_p, C.c_double]),
    "miosqp_qp_pool_launch": (C.c_int, [C.c_void_p, C.c_int32]),
    "miosqp_qp_pool_collect": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PoolDigest), C.c_int32, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32), i64p]),
    "miosqp_qp_cleanup": (C.c_int, [C.c_void_p]),
    "miosqp_qp_last_error": (C.c_char_p, []),
    "miosqp_qp_debug_iterate": (C.c_int, [C.c_void_p, C.c_int32, dp, dp, dp]),
    "miosqp_qp_get_scaling": (C.c_int, [C.c_void_p, dp, dp, dp]),
    "miosqp_qp_get_factor_stats": (C.c_int, [C.c_void_p, i64p]),
    "miosqp_qp_get_inverse_guard": (C.c_int, [C.c_void_p, dp]),
    "miosqp_qp_get_rho": (C.c_int, [C.c_void_p, dp]),
    "miosqp_qp_get_loop_stats": (C.c_int, [C.c_void_p, dp, i64p, C.c_int32]),
    "miosqp_qp_get_node_stats": (C.c_int, [C.c_void_p, dp, ip]),
    "miosqp_qp_get_loop_launches": (C.c_int, [C.c_void_p, i64p]),
    "miosqp_qp_get_batch_stats": (C.c_int, [C.c_void_p, dp, i64p, i64p, C.c_int32]),
    "miosqp_qp_debug_counter": (C.c_int64, [C.c_void_p, C.c_int32]),
    "miosqp_qp_debug_timeline": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32,
                                           C.POINTER(C.c_int32)]),
    "miosqp_qp_debug_clock": (C.c_int, [C.c_void_p, dp, dp]),
    "miosqp_qp_time_kernel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, dp, dp]),
}

_LIB = None


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C miosqp_amd/csrc` (there is no CPU fallback)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def last_error():
    msg = load().miosqp_qp_last_error()
    return msg.decode() if msg else ""


def as_d(a):
    return a.ctypes.data_as(dp)


def as_i(a):
    return a.ctypes.data_as(ip)


def source_digest():
    """16 hex digits identifying the device code (csrc/*.hip, *.inc, *.cpp, *.hpp): written into the PMC summaries under
    profiles/ by tools/pmc_merge.py and compared by bench.py, so that counter traffic of other code is not mixed with
    today's timing."""
    import glob
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_PKG, "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*"))):
        if f.endswith((".hip", ".inc", ".cpp", ".hpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
