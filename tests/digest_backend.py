"""Test double with the FULL surface of miosqp_amd.qp (solve_node, solve_batch, set_root, node
digest) built on the CPU oracle, so that the fused / batched / digest branches of the host layer
(miosqp_amd/bnb.py, miosqp_amd/dist.py) are exercised without a GPU.  The digest is computed with
the reference's own numpy expressions (workspace.py:232-272)."""
import types

import numpy as np

from oracle import oracle

constant = oracle.constant


class OSQP(oracle.OSQP):
    def setup(self, P=None, q=None, A=None, l=None, u=None, **kw):
        kw = {k: v for k, v in kw.items() if k not in ("max_batch", "fold", "resident", "device", "setup_on_device", "coop")}
        self._P, self._q, self._A = P.tocsc(), np.array(q, dtype=float), A.tocsc()
        self._root = None
        oracle.OSQP.setup(self, P, q, A, l, u, **kw)

    def update(self, q=None, l=None, u=None):
        if q is not None:
            self._q = np.array(q, dtype=float)
        oracle.OSQP.update(self, q=q, l=l, u=u)

    def set_integer_rows(self, i_idx, m_orig):
        self._ii, self._m = np.asarray(i_idx), int(m_orig)

    def set_root(self, l_root, u_root, eps_int_feas, eps_lin):
        self._root = (np.array(l_root, dtype=float), np.array(u_root, dtype=float), eps_int_feas, eps_lin)

    def solve_node(self, l, u, x0, y0):
        self.update(l=l, u=u)
        self.warm_start(x=x0, y=y0)
        r = self.solve()
        st, lower, digest = r.info.status_val, None, None
        x = r.x
        if st in (1, -2):
            k = len(self._ii)
            x[self._ii] = np.minimum(np.maximum(x[self._ii], np.asarray(l)[-k:]), np.asarray(u)[-k:])
            lower = .5 * np.dot(x, self._P.dot(x)) + np.dot(self._q, x)
            if self._root is not None:
                lr, ur, eps_int, eps_lin = self._root
                xi = x[self._ii]
                frac = abs(xi - np.round(xi))
                xr = np.copy(x)
                xr[self._ii] = np.round(xi)
                z = self._A.dot(xr)
                feas = not (np.any(z < lr - eps_lin) or np.any(z > ur + eps_lin))
                digest = types.SimpleNamespace(int_inf=int(np.sum(frac > eps_int)), nextvar=int(np.argmax(frac)),
                                               heur_feasible=feas,
                                               heur_obj=.5 * np.dot(xr, self._P.dot(xr)) + np.dot(self._q, xr))
        return types.SimpleNamespace(x=x, y=r.y, status_val=st, iter=r.info.iter, run_time=r.info.run_time,
                                     lower=lower, digest=digest)

    def solve_batch(self, l, u, x0, y0):
        rs = [self.solve_node(l[k], u[k], x0[k], y0[k]) for k in range(len(l))]
        return types.SimpleNamespace(
            x=np.stack([r.x for r in rs]), y=np.stack([r.y for r in rs]),
            status_val=np.array([r.status_val for r in rs]), iter=np.array([r.iter for r in rs]),
            lower=np.array([np.nan if r.lower is None else r.lower for r in rs]),
            run_time=np.array([r.run_time for r in rs]), digest=[r.digest for r in rs])
