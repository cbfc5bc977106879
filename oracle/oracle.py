"""ctypes wrapper around oracle/liboracle_qp.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It presents the same four-call surface the reference uses on its `osqp.OSQP` object
(/root/reference/miosqp/workspace.py:63-68, /root/reference/miosqp/node.py:102-125) so the
reference's own branch-and-bound layer can be driven with it when generating golden traces.
"""
import ctypes as C
import os
import subprocess
import types

import numpy as np
import scipy.sparse as spa

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Settings(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("rho", "sigma", "alpha", "eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf")] + \
               [(k, C.c_int) for k in ("max_iter", "scaling", "check_termination", "warm_start", "rho_auto")]


class Info(C.Structure):
    _fields_ = [("status_val", C.c_int), ("iter", C.c_int), ("obj_val", C.c_double),
                ("pri_res", C.c_double), ("dua_res", C.c_double), ("run_time", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle_qp.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.oqp_setup.restype = vp
        L.oqp_setup.argtypes = [C.c_int, C.c_int, ip, ip, dp, ip, ip, dp, dp, dp, dp,
                                C.POINTER(Settings)]
        L.oqp_default_settings.argtypes = [C.POINTER(Settings)]
        L.oqp_update_bounds.argtypes = [vp, dp, dp]
        L.oqp_update_lin_cost.argtypes = [vp, dp]
        L.oqp_warm_start.argtypes = [vp, dp, dp]
        L.oqp_solve.argtypes = [vp, dp, dp, C.POINTER(Info)]
        L.oqp_iterate.argtypes = [vp, C.c_int]
        L.oqp_get_iterates.argtypes = [vp, dp, dp, dp]
        L.oqp_get_scaling.argtypes = [vp, dp, dp, dp]
        L.oqp_factor_nnz.argtypes = [vp]
        L.oqp_get_rho.argtypes = [vp]
        L.oqp_get_rho.restype = C.c_double
        L.oqp_cleanup.argtypes = [vp]
        L.oqp_cleanup.restype = None
        L.oqp_constant.argtypes = [C.c_char_p]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def constant(name):
    return lib().oqp_constant(name.encode())


_ALIASES = {"eps_inf": "eps_prim_inf", "early_terminate_interval": "check_termination"}
_IGNORED = {"verbose", "polish", "polishing", "adaptive_rho", "linsys_solver", "time_limit",
            "scaled_termination", "eps_unb"}


def make_settings(kw):
    s = Settings()
    lib().oqp_default_settings(C.byref(s))
    for k, v in kw.items():
        k = _ALIASES.get(k, k)
        if k == "adaptive_rho" and v:
            raise ValueError("adaptive_rho is not part of the frozen spec (DESIGN.md)")
        if k in _IGNORED:
            continue
        if k == "rho" and isinstance(v, str):
            if v != "auto":
                raise ValueError("rho: a number or 'auto'")
            s.rho_auto = 1  # chosen once at setup from the default starting value, then frozen
            continue
        if not hasattr(s, k):
            raise TypeError("unknown setting %r" % k)
        setattr(s, k, v)
    return s


class OSQP(object):
    """Same method surface as the `osqp.OSQP` object the reference drives."""

    def __init__(self):
        self._h = None

    def setup(self, P=None, q=None, A=None, l=None, u=None, **settings):
        L = lib()
        P = spa.csc_matrix(P)
        A = spa.csc_matrix(A)
        P.sort_indices()
        A.sort_indices()
        self.n, self.m = A.shape[1], A.shape[0]
        s = make_settings(settings)
        self._keep = [np.ascontiguousarray(v, dtype=t) for v, t in (
            (P.indptr, np.int32), (P.indices, np.int32), (P.data, np.float64),
            (A.indptr, np.int32), (A.indices, np.int32), (A.data, np.float64),
            (q, np.float64), (l, np.float64), (u, np.float64))]
        k = self._keep
        self._h = L.oqp_setup(self.n, self.m, _i(k[0]), _i(k[1]), _d(k[2]), _i(k[3]), _i(k[4]),
                              _d(k[5]), _d(k[6]), _d(k[7]), _d(k[8]), C.byref(s))
        if not self._h:
            raise ValueError("oracle setup failed (l > u or singular KKT)")

    def update(self, q=None, l=None, u=None):
        L = lib()
        if q is not None:
            q = np.ascontiguousarray(q, dtype=np.float64)
            L.oqp_update_lin_cost(self._h, _d(q))
        if l is not None or u is not None:
            l = np.ascontiguousarray(l, dtype=np.float64)
            u = np.ascontiguousarray(u, dtype=np.float64)
            if L.oqp_update_bounds(self._h, _d(l), _d(u)):
                raise ValueError("lower bound must be lower than or equal to upper bound")

    def warm_start(self, x=None, y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        lib().oqp_warm_start(self._h, _d(x), _d(y))

    def solve(self):
        x = np.empty(self.n)
        y = np.empty(self.m)
        info = Info()
        lib().oqp_solve(self._h, _d(x), _d(y), C.byref(info))
        return types.SimpleNamespace(x=x, y=y, info=info)

    # parity hooks
    def iterate(self, k):
        lib().oqp_iterate(self._h, k)

    def iterates(self):
        x, z, y = np.empty(self.n), np.empty(self.m), np.empty(self.m)
        lib().oqp_get_iterates(self._h, _d(x), _d(z), _d(y))
        return x, z, y

    def scaling(self):
        D, E, c = np.empty(self.n), np.empty(self.m), C.c_double()
        lib().oqp_get_scaling(self._h, _d(D), _d(E), C.byref(c))
        return D, E, c.value

    def factor_nnz(self):
        return lib().oqp_factor_nnz(self._h)

    def rho(self):
        """the rho in use (differs from the setting after rho="auto")"""
        return lib().oqp_get_rho(self._h)

    def __del__(self):
        if self._h and _LIB is not None:
            _LIB.oqp_cleanup(self._h)
            self._h = None
