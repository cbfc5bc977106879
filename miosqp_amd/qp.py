"""`osqp`-shaped front of the HIP relaxation engine.

Presents exactly the surface the reference's branch-and-bound layer uses on its `osqp` module
(SURVEY.md sec. 8b):

    osqp.OSQP().setup(P, q, A, l, u, **qp_settings)      /root/reference/miosqp/workspace.py:63-68
    solver.update(l=, u=) / solver.update(q=)            node.py:102 / solver.py:185
    solver.warm_start(x=, y=)                            node.py:105
    solver.solve() -> .x .y .info.status_val .iter .run_time          node.py:108-125
    osqp.constant(name)                                  node.py:88,128-129

plus the fused / batched extensions `solve_node`, `solve_batch`.  Everything numeric happens in
libmiosqp_hip.so through the C ABI of include/miosqp_amd.h; this file only marshals arrays.
"""
import ctypes as C
import types

import numpy as np
import scipy.sparse as spa

from miosqp_amd import _lib

# accepted spellings from older OSQP releases (the reference's own fixtures use them:
# /root/reference/max_iter_examples/*.pickle carry 'eps_inf' and 'polishing')
_ALIASES = {"eps_inf": "eps_prim_inf", "eps_unb": "eps_dual_inf",
            "early_terminate_interval": "check_termination"}
# accepted and ignored: no effect on the iteration
_IGNORED = {"verbose", "polish", "polishing", "linsys_solver", "time_limit", "scaled_termination",
            "delta", "polish_refine_iter", "early_terminate"}


def constant(name):
    """osqp.constant(name)."""
    v = _lib.load().miosqp_qp_constant(name.encode())
    if v == 0:
        raise ValueError("Constant %s not found" % name)
    return v


def default_settings():
    s = _lib.Settings()
    _lib.load().miosqp_qp_default_settings(C.byref(s))
    return s


def _settings_from_kwargs(kw):
    s = default_settings()
    for k, v in kw.items():
        k = _ALIASES.get(k, k)
        if k in _IGNORED:
            continue
        if k == "adaptive_rho":
            if v:
                raise ValueError("adaptive_rho is not supported: rho is one fixed scalar so that "
                                 "the KKT factor is shared by every node (DESIGN.md)")
            continue
        if k == "rho" and isinstance(v, str):
            if v != "auto":
                raise ValueError("rho: a number or 'auto'")
            s.rho_auto = 1  # chosen once at setup (miosqp_qp_settings.rho_auto), from the default starting value
            continue
        if k not in dict(s._fields_) or k == "reserved":
            raise TypeError("setup() got an unexpected setting %r" % k)
        setattr(s, k, v)
    return s


def _check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, _lib.last_error()))
    return rc


def _f64(a, size, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != (size,):
        raise ValueError("%s must have %d entries" % (name, size))
    return a


class DevicePtr(object):
    """`count` doubles at address `addr` of DEVICE memory (a slice of a torch tensor, say): accepted wherever a leaf's
    vectors go in or come out (search_add_leaf / search_take_leaf, stream_*, pool_write_node / pool_read_node) -- the
    library copies device to device, a leaf moves between ranks without visiting the host (dist.ShardedStream)."""

    def __init__(self, addr, count):
        self.addr, self.count = int(addr), int(count)

    def __len__(self):
        return self.count


def leaf_record_views(tensor, n_int, n, m):
    """DevicePtr views (l_int, u_int, x0, y0) of a device tensor of >= 2 n_int + n + m doubles"""
    base = tensor.data_ptr()
    offs = (0, n_int, 2 * n_int, 2 * n_int + n)
    cnts = (n_int, n_int, n, m)
    return tuple(DevicePtr(base + 8 * o, c) for o, c in zip(offs, cnts))


def _vec(a, size, name):
    """ctypes pointer of a leaf vector given as an array (host) or a DevicePtr"""
    if isinstance(a, DevicePtr):
        if a.count != size:
            raise ValueError("%s must have %d entries" % (name, size))
        return C.cast(C.c_void_p(a.addr), _lib.dp)
    return _lib.as_d(_f64(a, size, name))


class OSQP(object):
    def __init__(self):
        self._h = None
        self._lib = _lib.load()
        self.n = self.m = 0

    # -- setup -----------------------------------------------------------------------------
    def setup(self, P=None, q=None, A=None, l=None, u=None, **settings):
        if self._h is not None:
            raise RuntimeError("setup() called twice")
        P = spa.csc_matrix(P)
        A = spa.csc_matrix(A)
        P.sort_indices()
        A.sort_indices()
        n, m = A.shape[1], A.shape[0]
        if P.shape != (n, n):
            raise ValueError("P must be %d x %d" % (n, n))
        s = _settings_from_kwargs(settings)
        arrs = [np.ascontiguousarray(P.indptr, dtype=np.int32),
                np.ascontiguousarray(P.indices, dtype=np.int32),
                np.ascontiguousarray(P.data, dtype=np.float64),
                np.ascontiguousarray(A.indptr, dtype=np.int32),
                np.ascontiguousarray(A.indices, dtype=np.int32),
                np.ascontiguousarray(A.data, dtype=np.float64),
                _f64(q, n, "q"), _f64(l, m, "l"), _f64(u, m, "u")]
        if np.any(arrs[7] > arrs[8]):
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        h = C.c_void_p()
        rc = self._lib.miosqp_qp_setup(C.byref(h), n, m, _lib.as_i(arrs[0]), _lib.as_i(arrs[1]),
                                       _lib.as_d(arrs[2]), _lib.as_i(arrs[3]), _lib.as_i(arrs[4]),
                                       _lib.as_d(arrs[5]), _lib.as_d(arrs[6]), _lib.as_d(arrs[7]),
                                       _lib.as_d(arrs[8]), C.byref(s))
        _check(rc, "setup")
        self._h, self.n, self.m, self.settings = h, n, m, s

    def set_integer_rows(self, i_idx, m_orig):
        ii = np.ascontiguousarray(i_idx, dtype=np.int32)
        _check(self._lib.miosqp_qp_set_integer_rows(self._h, len(ii), _lib.as_i(ii), int(m_orig)),
               "set_integer_rows")

    def set_root(self, l_root, u_root, eps_int_feas, eps_lin):
        """Enable the on-device node digest (see miosqp_qp_set_root)."""
        l_root, u_root = _f64(l_root, self.m, "l_root"), _f64(u_root, self.m, "u_root")
        _check(self._lib.miosqp_qp_set_root(self._h, _lib.as_d(l_root), _lib.as_d(u_root),
                                            float(eps_int_feas), float(eps_lin)), "set_root")

    # -- the reference's four calls ------------------------------------------------------------
    def update(self, q=None, l=None, u=None):
        if q is not None:
            _check(self._lib.miosqp_qp_update_lin_cost(self._h, _lib.as_d(_f64(q, self.n, "q"))),
                   "update(q)")
        if l is not None or u is not None:
            if l is None or u is None:
                raise ValueError("update() needs both l and u")
            l, u = _f64(l, self.m, "l"), _f64(u, self.m, "u")
            rc = _check(self._lib.miosqp_qp_update_bounds(self._h, _lib.as_d(l), _lib.as_d(u)),
                        "update(l,u)")
            if rc == 1:
                raise ValueError("Lower bound must be lower than or equal to upper bound")

    def warm_start(self, x=None, y=None):
        if x is None or y is None:
            raise ValueError("warm_start() needs both x and y")
        _check(self._lib.miosqp_qp_warm_start(self._h, _lib.as_d(_f64(x, self.n, "x")),
                                              _lib.as_d(_f64(y, self.m, "y"))), "warm_start")

    def solve(self):
        x, y, info = np.empty(self.n), np.empty(self.m), _lib.Info()
        _check(self._lib.miosqp_qp_solve(self._h, _lib.as_d(x), _lib.as_d(y), C.byref(info)), "solve")
        return types.SimpleNamespace(x=x, y=y, info=info)

    # -- fused / batched extensions ------------------------------------------------------------
    def solve_node(self, l, u, x0, y0):
        """Whole body of Node.solve() (node.py:96-143) in one device round trip."""
        l, u = _f64(l, self.m, "l"), _f64(u, self.m, "u")
        x0, y0 = _f64(x0, self.n, "x0"), _f64(y0, self.m, "y0")
        x, y, info = np.empty(self.n), np.empty(self.m), _lib.Info()
        rc = _check(self._lib.miosqp_qp_solve_node(self._h, _lib.as_d(l), _lib.as_d(u), _lib.as_d(x0),
                                                   _lib.as_d(y0), _lib.as_d(x), _lib.as_d(y),
                                                   C.byref(info)), "solve_node")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        lower = None if np.isnan(info.lower) else info.lower
        digest = None
        if info.int_inf >= 0:
            digest = types.SimpleNamespace(int_inf=info.int_inf, nextvar=info.nextvar,
                                           heur_feasible=info.heur_viol <= 0.0, heur_obj=info.heur_obj,
                                           info_viol=info.heur_viol)
        return types.SimpleNamespace(x=x, y=y, status_val=info.status_val, iter=info.iter,
                                     run_time=info.run_time, lower=lower, info=info, digest=digest)

    def solve_batch(self, l, u, x0, y0):
        """B independent nodes sharing the factor; arrays are [B, .]."""
        l = np.ascontiguousarray(l, dtype=np.float64)
        B = l.shape[0]
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(B, self.m)
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.n)
        y0 = np.ascontiguousarray(y0, dtype=np.float64).reshape(B, self.m)
        x, y = np.empty((B, self.n)), np.empty((B, self.m))
        infos = (_lib.Info * B)()
        rc = _check(self._lib.miosqp_qp_solve_batch(self._h, B, _lib.as_d(l), _lib.as_d(u),
                                                    _lib.as_d(x0), _lib.as_d(y0), _lib.as_d(x),
                                                    _lib.as_d(y), infos), "solve_batch")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        digests = [None] * B
        for k, i in enumerate(infos):
            if i.int_inf >= 0:
                digests[k] = types.SimpleNamespace(int_inf=i.int_inf, nextvar=i.nextvar,
                                                   heur_feasible=i.heur_viol <= 0.0, heur_obj=i.heur_obj,
                                                   info_viol=i.heur_viol)
        return types.SimpleNamespace(
            digest=digests,
            x=x, y=y, status_val=np.array([i.status_val for i in infos]),
            iter=np.array([i.iter for i in infos]), lower=np.array([i.lower for i in infos]),
            run_time=np.array([i.run_time for i in infos]), infos=infos)

    def solve_tree(self, l, u, x0, y0, upper0, x_inc0, tree_explor_rule, max_iter_bb):
        """A whole tree search in one launch (small problems); None when the engine does not cover this size."""
        l, u = _f64(l, self.m, "l"), _f64(u, self.m, "u")
        x0, y0 = _f64(x0, self.n, "x0"), _f64(y0, self.m, "y0")
        have = x_inc0 is not None and np.isfinite(upper0)
        xin = _f64(x_inc0, self.n, "x_inc0") if have else None
        x, info = np.empty(self.n), _lib.TreeInfo()
        # the launch counts nodes in 32 bits; the reference accepts any number here (also inf): clamp
        max_iter_bb = 2 ** 31 - 1 if not np.isfinite(max_iter_bb) else int(min(int(max_iter_bb), 2 ** 31 - 1))
        rc = self._lib.miosqp_qp_solve_tree(self._h, _lib.as_d(l), _lib.as_d(u), _lib.as_d(x0), _lib.as_d(y0),
                                            float(upper0) if have else float("inf"), _lib.as_d(xin) if have else None,
                                            int(tree_explor_rule), int(max_iter_bb), _lib.as_d(x), C.byref(info))
        if rc == -5:
            return None
        _check(rc, "solve_tree")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        return types.SimpleNamespace(x=x, info=info)

    def solve_trees(self, q, l, u, x0, y0, upper0, x_inc0, tree_explor_rule, max_iter_bb):
        """B MIQPs on this engine's factor (q, l, u, x0, y0 instance-major: B x n / B x M), every tree in one launch.
        upper0: B incumbent values (inf: none), x_inc0: B x n or None.  Returns (x B x n, [TreeInfo]) or None when the
        engine does not cover this size."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        B = q.shape[0]
        l, u = np.ascontiguousarray(l, dtype=np.float64), np.ascontiguousarray(u, dtype=np.float64)
        x0, y0 = np.ascontiguousarray(x0, dtype=np.float64), np.ascontiguousarray(y0, dtype=np.float64)
        if q.shape != (B, self.n) or x0.shape != (B, self.n) or l.shape != (B, self.m) or u.shape != (B, self.m) or y0.shape != (B, self.m):
            raise ValueError("solve_trees: instance-major arrays (B x n, B x M)")
        up = np.minimum(np.ascontiguousarray(upper0, dtype=np.float64).reshape(B), 1.7e308)
        xin = None if x_inc0 is None else np.ascontiguousarray(x_inc0, dtype=np.float64).reshape(B, self.n)
        x = np.zeros((B, self.n))
        infos = (_lib.TreeInfo * B)()
        max_iter_bb = 2 ** 31 - 1 if not np.isfinite(max_iter_bb) else int(min(int(max_iter_bb), 2 ** 31 - 1))
        rc = self._lib.miosqp_qp_solve_trees(self._h, B, _lib.as_d(q), _lib.as_d(l), _lib.as_d(u), _lib.as_d(x0), _lib.as_d(y0),
                                             _lib.as_d(up), None if xin is None else _lib.as_d(xin), int(tree_explor_rule),
                                             max_iter_bb, _lib.as_d(x), infos)
        if rc == -5:
            return None
        _check(rc, "solve_trees")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")
        return x, list(infos)

    # -- node-at-a-time branch and bound driven from the host in C++ (miosqp_qp_search_*) --------------
    def search_create(self, capacity):
        _check(self._lib.miosqp_qp_search_create(self._h, int(capacity)), "search_create")
        self._search_info = _lib.SearchInfo()

    def search_reset(self):
        _check(self._lib.miosqp_qp_search_reset(self._h), "search_reset")

    def search_add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        k = len(l_int)
        rc = _check(self._lib.miosqp_qp_search_add_leaf(
            self._h, _vec(l_int, k, "l_int"), _vec(u_int, k, "u_int"), _vec(x0, self.n, "x0"), _vec(y0, self.m, "y0"),
            int(depth), float(lower)), "search_add_leaf")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")

    def search_take_leaf(self, n_int, into=None):
        """(l_int, u_int, x0, y0, depth, lower) of the shallowest open leaf, removed from the list; None if none.
        into: four DevicePtr (l_int, u_int, x0, y0) the vectors are copied to instead (device to device)."""
        l, u, x, y = into if into is not None else (np.empty(n_int), np.empty(n_int), np.empty(self.n), np.empty(self.m))
        depth, lower = C.c_int32(), C.c_double()
        rc = _check(self._lib.miosqp_qp_search_take_leaf(self._h, _vec(l, n_int, "l_int"), _vec(u, n_int, "u_int"),
                                                         _vec(x, self.n, "x0"), _vec(y, self.m, "y0"),
                                                         C.byref(depth), C.byref(lower)), "search_take_leaf")
        return None if rc == 1 else (l, u, x, y, depth.value, lower.value)

    def search_set_incumbent(self, upper, x):
        """x None: only the VALUE of the incumbent the search already holds is replaced (a rounding-heuristic value
        recomputed on the host, see SearchInfo.improved == 2)."""
        _check(self._lib.miosqp_qp_search_set_incumbent(self._h, float(upper),
                                                        None if x is None else _lib.as_d(_f64(x, self.n, "x"))),
               "search_set_incumbent")

    def search_get_incumbent(self):
        """(upper, x); x is None while there is no incumbent."""
        upper, x = C.c_double(), np.empty(self.n)
        _check(self._lib.miosqp_qp_search_get_incumbent(self._h, C.byref(upper), _lib.as_d(x)), "search_get_incumbent")
        return (upper.value, x) if upper.value < 1.7e308 else (float("inf"), None)

    def search_run(self, tree_explor_rule, max_nodes, budget_s=0.0):
        """Returns the info record; `info.full` is set when the slot store could not grow any further (the device is
        out of memory): the counters of the nodes done in this call are valid, the caller decides what to do."""
        info = self._search_info
        max_nodes = 2 ** 62 if not np.isfinite(max_nodes) else int(min(int(max_nodes), 2 ** 62))
        rc = self._lib.miosqp_qp_search_run(self._h, int(tree_explor_rule), max_nodes, float(budget_s), C.byref(info))
        info.full = rc == -6
        if not info.full:
            _check(rc, "search_run")
        return info

    # -- the host side of the streaming search, compiled (miosqp_qp_stream_*) ---------------------------
    def stream_create(self, capacity, columns, ring_margin=0):
        _check(self._lib.miosqp_qp_stream_create(self._h, int(capacity), int(columns), int(ring_margin)), "stream_create")
        self._stream_info = _lib.StreamInfo()

    def stream_begin(self):
        _check(self._lib.miosqp_qp_stream_begin(self._h), "stream_begin")

    def stream_add_leaf(self, l_int, u_int, x0, y0, depth, lower):
        k = len(l_int)
        rc = _check(self._lib.miosqp_qp_stream_add_leaf(
            self._h, _vec(l_int, k, "l_int"), _vec(u_int, k, "u_int"), _vec(x0, self.n, "x0"), _vec(y0, self.m, "y0"),
            int(depth), float(lower)), "stream_add_leaf")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")

    def stream_take_leaf(self, n_int, into=None):
        l, u, x, y = into if into is not None else (np.empty(n_int), np.empty(n_int), np.empty(self.n), np.empty(self.m))
        depth, lower = C.c_int32(), C.c_double()
        rc = _check(self._lib.miosqp_qp_stream_take_leaf(self._h, _vec(l, n_int, "l_int"), _vec(u, n_int, "u_int"),
                                                         _vec(x, self.n, "x0"), _vec(y, self.m, "y0"),
                                                         C.byref(depth), C.byref(lower)), "stream_take_leaf")
        return None if rc == 1 else (l, u, x, y, depth.value, lower.value)

    def stream_set_incumbent(self, upper, x):
        _check(self._lib.miosqp_qp_stream_set_incumbent(self._h, float(min(upper, 1.7e308)),
                                                        _lib.as_d(_f64(x, self.n, "x"))), "stream_set_incumbent")

    def stream_get_incumbent(self):
        upper, x = C.c_double(), np.empty(self.n)
        _check(self._lib.miosqp_qp_stream_get_incumbent(self._h, C.byref(upper), _lib.as_d(x)), "stream_get_incumbent")
        return (upper.value, x) if upper.value < 1.7e308 else (float("inf"), None)

    def stream_step(self, tree_explor_rule, chunks=1, rounds=1, max_nodes=0):
        info = self._stream_info
        _check(self._lib.miosqp_qp_stream_step(self._h, int(tree_explor_rule), int(chunks), int(rounds), int(max_nodes),
                                               C.byref(info)), "stream_step")
        return info

    # -- device-resident leaf pool + streaming batch (miosqp_qp_pool_*) ------------------------------
    POOL_PRUNED = -100

    def pool_create(self, capacity, columns):
        _check(self._lib.miosqp_qp_pool_create(self._h, int(capacity), int(columns)), "pool_create")

    def _digest_buffer(self):
        if getattr(self, "_dg", None) is None:
            self._dg = (_lib.PoolDigest * 8192)()
            self._dg_np = np.frombuffer(self._dg, dtype=np.dtype(
                [("slot", "i4"), ("status_val", "i4"), ("iter", "i4"), ("int_inf", "i4"), ("nextvar", "i4"),
                 ("reserved", "i4"), ("lower", "f8"), ("heur_viol", "f8"), ("heur_obj", "f8"), ("pri_res", "f8"),
                 ("dua_res", "f8")]))
        return self._dg

    def pool_reset(self):
        _check(self._lib.miosqp_qp_pool_reset(self._h), "pool_reset")

    def pool_write_node(self, slot, l_int, u_int, x0, y0):
        k = len(l_int)
        rc = _check(self._lib.miosqp_qp_pool_write_node(
            self._h, int(slot), _vec(l_int, k, "l_int"), _vec(u_int, k, "u_int"), _vec(x0, self.n, "x0"),
            _vec(y0, self.m, "y0")), "pool_write_node")
        if rc == 1:
            raise ValueError("Lower bound must be lower than or equal to upper bound")

    def pool_read_node(self, slot, n_int, want=("l", "u", "x", "y"), into=None):
        """into: dict of DevicePtr by the same keys: those vectors are copied there (device to device) instead"""
        out = dict(l=np.empty(n_int), u=np.empty(n_int), x=np.empty(self.n), y=np.empty(self.m))
        if into:
            out.update(into)
        size = dict(l=n_int, u=n_int, x=self.n, y=self.m)
        ptr = [(_vec(out[k], size[k], k) if k in want else None) for k in ("l", "u", "x", "y")]
        _check(self._lib.miosqp_qp_pool_read_node(self._h, int(slot), *ptr), "pool_read_node")
        return types.SimpleNamespace(**{k: out[k] for k in want})

    def pool_push(self, slot, child0, child1, lower):
        s = np.ascontiguousarray(slot, dtype=np.int32)
        c0 = np.ascontiguousarray(child0, dtype=np.int32)
        c1 = np.ascontiguousarray(child1, dtype=np.int32)
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        _check(self._lib.miosqp_qp_pool_push(self._h, len(s), _lib.as_i(s), _lib.as_i(c0), _lib.as_i(c1),
                                             _lib.as_d(lo)), "pool_push")

    def pool_set_upper(self, upper):
        _check(self._lib.miosqp_qp_pool_set_upper(self._h, float(min(upper, 1.7e308))), "pool_set_upper")

    def pool_launch(self, chunks=1):
        _check(self._lib.miosqp_qp_pool_launch(self._h, int(chunks)), "pool_launch")

    def pool_collect(self, keep_in_flight=0):
        """Waits until at most `keep_in_flight` launches are still running; returns (digests as a numpy record
        array (a copy), active columns, ready-ring entries not yet taken)."""
        n, act, left = C.c_int32(), C.c_int32(), C.c_int64()
        _check(self._lib.miosqp_qp_pool_collect(self._h, int(keep_in_flight), self._digest_buffer(), 8192, C.byref(n),
                                                C.byref(act), C.byref(left)), "pool_collect")
        return self._dg_np[:n.value].copy(), act.value, left.value

    # -- introspection ------------------------------------------------------------------------
    def debug_iterate(self, k):
        x, z, y = np.empty(self.n), np.empty(self.m), np.empty(self.m)
        _check(self._lib.miosqp_qp_debug_iterate(self._h, k, _lib.as_d(x), _lib.as_d(z), _lib.as_d(y)),
               "debug_iterate")
        return x, z, y

    def scaling(self):
        D, E, c = np.empty(self.n), np.empty(self.m), C.c_double()
        _check(self._lib.miosqp_qp_get_scaling(self._h, _lib.as_d(D), _lib.as_d(E), C.byref(c)),
               "get_scaling")
        return D, E, c.value

    def factor_stats(self):
        out = np.zeros(10, dtype=np.int64)
        _check(self._lib.miosqp_qp_get_factor_stats(self._h, out.ctypes.data_as(_lib.i64p)),
               "get_factor_stats")
        return dict(nnz_L=int(out[0]), nnz_panel=int(out[1]), tail_order=int(out[2]),
                    bytes_per_iter=int(out[3]), bytes_moved_per_iter=int(out[8]), coop_fallbacks=int(out[9]),
                    tpr=(int(out[4]), int(out[5]), int(out[6])),
                    fold=bool(out[7] & 1), resident=bool(out[7] & 2), setup_on_device=bool(out[7] & 4), coop=bool(out[7] & 8), pers=bool(out[7] & 16), tail_inverse=bool(out[7] & 32), pers_small=bool(out[7] & 64), coop_nap=(out[7] >> 8) & 0xff,
                    batch_pers=bool(out[7] & (1 << 16)), inverse_guard_tripped=bool(out[7] & (1 << 17)),
                    search_grid_resident=bool(out[7] & (1 << 18)))

    def rho(self):
        """the rho in use (differs from the setting after rho="auto")"""
        out = np.zeros(1)
        _check(self._lib.miosqp_qp_get_rho(self._h, out.ctypes.data_as(_lib.dp)), "get_rho")
        return float(out[0])

    def inverse_guard(self):
        """(residual, threshold, tripped) of the set-up check of the explicit KKT inverse; residual -1: none was built."""
        out = np.zeros(3)
        _check(self._lib.miosqp_qp_get_inverse_guard(self._h, out.ctypes.data_as(_lib.dp)), "get_inverse_guard")
        return float(out[0]), float(out[1]), bool(out[2])

    def loop_stats(self, reset=False):
        ms, it = C.c_double(), C.c_int64()
        _check(self._lib.miosqp_qp_get_loop_stats(self._h, C.byref(ms), C.byref(it), int(reset)),
               "get_loop_stats")
        return ms.value, it.value

    def node_stats(self):
        """(min, median, max) microseconds of device time per ADMM iteration over the nodes of the hosted search since the
        last loop_stats(reset=True), and the number of nodes."""
        v, k = np.zeros(3), C.c_int32()
        _check(self._lib.miosqp_qp_get_node_stats(self._h, _lib.as_d(v), C.byref(k)), "get_node_stats")
        return float(v[0]), float(v[1]), float(v[2]), k.value

    def loop_launches(self):
        """Launches of the hosted search's solver kernel since the last loop_stats(reset=True): one per node, or one per
        search_run where the cooperative grid stays resident (k_coop_run)."""
        k = C.c_int64()
        _check(self._lib.miosqp_qp_get_loop_launches(self._h, C.byref(k)), "get_loop_launches")
        return k.value

    def batch_stats(self, reset=False):
        ms, bi, ni = C.c_double(), C.c_int64(), C.c_int64()
        _check(self._lib.miosqp_qp_get_batch_stats(self._h, C.byref(ms), C.byref(bi), C.byref(ni), int(reset)),
               "get_batch_stats")
        return ms.value, bi.value, ni.value

    def compactions(self):
        return int(self._lib.miosqp_qp_debug_counter(self._h, 0))

    def batch_pers_fallbacks(self):
        """Times a chunk's persistent launch (kbp) was called off and the engine went back to the launches."""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 1))

    def stream_chunks_by_form(self):
        """(launches of the stream's persistent kernel kbs, chunks queued through them, chunks queued as the chunk graph)"""
        return tuple(int(self._lib.miosqp_qp_debug_counter(self._h, k)) for k in (7, 8, 9))

    def resident_grid_poll_delay(self):
        """(poll delay the resident search grid ran with last, or -1; synthetic nodes its calibration ran on this engine)"""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 10)), int(self._lib.miosqp_qp_debug_counter(self._h, 11))

    def chip_turn_waits(self):
        """Whole-chip launches on this engine's device that were ordered behind another engine's (they take turns)."""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 2))

    def chip_turn_users(self):
        return int(self._lib.miosqp_qp_debug_counter(self._h, 4))

    def tail_inverse_resident_columns(self):
        """Columns of every row of S^-1 the persistent streaming solver keeps in LDS for a whole launch (0: none)."""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 5))

    def tail_inverse_tiles(self):
        """Tiles of S^-1 (on and above the diagonal, one per workgroup) the persistent streaming solver reads per iteration
        when it takes the matrix as symmetric; 0: it streams whole rows."""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 6))

    def call_off_word(self):
        """The control block's call-off / time-out word once the engine's stream is idle (0: nothing happened)."""
        return int(self._lib.miosqp_qp_debug_counter(self._h, 3))

    def time_kernel(self, which, reps=200):
        us, by = C.c_double(), C.c_double()
        _check(self._lib.miosqp_qp_time_kernel(self._h, which, reps, C.byref(us), C.byref(by)),
               "time_kernel")
        return us.value, by.value

    def close(self):
        if self._h is not None:
            self._lib.miosqp_qp_cleanup(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
