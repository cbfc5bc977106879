"""Independent numpy fp64 checkers for QP solutions (no solver code shared with oracle/ or
miosqp_amd/).  Problem:  min .5 x'Px + q'x  s.t.  l <= Ax <= u."""
import numpy as np


def kkt_certificate(P, q, A, l, u, x, y):
    """Returns dict of unscaled optimality measures."""
    Ax = A.dot(x)
    pri = np.max(np.maximum(np.maximum(l - Ax, Ax - u), 0.0)) if len(l) else 0.0
    dua = np.max(np.abs(P.dot(x) + q + A.T.dot(y)))
    # complementarity: y_i > 0 only where Ax_i = u_i, y_i < 0 only where Ax_i = l_i
    yp, ym = np.maximum(y, 0), np.minimum(y, 0)
    fin_u, fin_l = np.isfinite(u), np.isfinite(l)
    comp = 0.0
    if fin_u.any():
        comp = max(comp, np.max(np.abs(yp[fin_u] * (u[fin_u] - Ax[fin_u]))))
    if fin_l.any():
        comp = max(comp, np.max(np.abs(ym[fin_l] * (Ax[fin_l] - l[fin_l]))))
    # multipliers on infinite sides must vanish
    stray = max(np.max(yp[~fin_u], initial=0.0), np.max(-ym[~fin_l], initial=0.0))
    return dict(pri=pri, dua=dua, comp=comp, stray=stray,
                obj=0.5 * x.dot(P.dot(x)) + q.dot(x))


def osqp_tolerances(P, q, A, x, y, z, eps_abs, eps_rel):
    """eps_prim, eps_dual of the OSQP termination rule (paper sec. 3.4), unscaled."""
    Ax = A.dot(x)
    eps_pri = eps_abs + eps_rel * max(np.max(np.abs(Ax), initial=0), np.max(np.abs(z), initial=0))
    eps_dua = eps_abs + eps_rel * max(np.max(np.abs(P.dot(x))), np.max(np.abs(A.T.dot(y)), initial=0),
                                      np.max(np.abs(q)))
    return eps_pri, eps_dua


def active_set_refine(P, q, A, l, u, x, y, tol=1e-7):
    """Solve the equality-constrained QP on the active set read off (x, y): an exact solution
    to compare an ADMM answer against (dense, small problems only)."""
    P = np.asarray(P.todense()) if hasattr(P, "todense") else np.asarray(P)
    A = np.asarray(A.todense()) if hasattr(A, "todense") else np.asarray(A)
    Ax = A.dot(x)
    scale = 1e-3 * max(1.0, np.max(np.abs(y)))
    up = (y > scale) | (np.isfinite(u) & (np.abs(Ax - u) < tol) & (y >= 0) & (l == u))
    lo = (y < -scale) & ~up
    act = up | lo
    b = np.where(up, u, l)[act]
    Aa = A[act]
    n, k = P.shape[0], Aa.shape[0]
    K = np.block([[P, Aa.T], [Aa, np.zeros((k, k))]])
    sol = np.linalg.lstsq(K, np.concatenate([-q, b]), rcond=None)[0]
    return sol[:n]
