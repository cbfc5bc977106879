import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from miosqp_amd import problems, qp
from oracle import oracle
from test_gpu_parity import _frontier

pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
m = pr["A"].shape[0]
def eng(cap):
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=cap))
    g.set_integer_rows(pr["i_idx"], m)
    g.set_root(l, u, 1e-3, 1e-3)
    return g
g = eng(256)
leaves = _frontier(g, pr, l, u, 256)[:256]
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
print("hash", hash(L.tobytes()) & 0xffff, hash(X.tobytes()) & 0xffff)
o = oracle.OSQP(); o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
def orc(k):
    o.update(l=L[k], u=U[k]); o.warm_start(x=X[k], y=Y[k]); r = o.solve(); return r.info.status_val, r.info.iter
rb = g.solve_batch(L, U, X, Y)
print("batch iters first 8", rb.iter[:8])
s = [g.solve_node(L[k], U[k], X[k], Y[k]) for k in range(8)]
print("node after batch   ", [r.iter for r in s])
print("oracle             ", [orc(k)[1] for k in range(8)])
g2 = eng(256)
s2 = [g2.solve_node(L[k], U[k], X[k], Y[k]) for k in range(8)]
print("fresh engine node  ", [r.iter for r in s2])
rb2 = g2.solve_batch(L, U, X, Y)
print("fresh engine batch ", rb2.iter[:8])
