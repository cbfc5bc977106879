"""Debug: config-3 wave, solve_batch vs solve_node vs oracle; which columns disagree and under which settings."""
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
singles = [g.solve_node(L[k], U[k], X[k], Y[k]) for k in range(256)]
it1 = np.array([r.iter for r in singles]); st1 = np.array([r.status_val for r in singles])
print("node iters: min %d max %d, statuses %s" % (it1.min(), it1.max(), np.unique(st1, return_counts=True)))
o = oracle.OSQP(); o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
def orc(k):
    o.update(l=L[k], u=U[k]); o.warm_start(x=X[k], y=Y[k]); r = o.solve(); return r.info.status_val, r.info.iter
for k in (0, 1, 2, 100, 255):
    print("col", k, "node", (singles[k].status_val, singles[k].iter), "oracle", orc(k))
for cap, compact, cnt in ((256, "1", 256), (256, "0", 256), (64, "1", 256), (256, "1", 64), (256, "1", 128), (256, "1", 65)):
    os.environ["MIOSQP_COMPACT"] = compact
    gg = eng(cap)
    rb = gg.solve_batch(L[:cnt], U[:cnt], X[:cnt], Y[:cnt])
    bad = [k for k in range(cnt) if (rb.status_val[k], rb.iter[k]) != (st1[k], it1[k])]
    print("cap %d compact %s count %d: %d mismatches; first %s" % (cap, compact, cnt, len(bad),
          [(k, int(rb.iter[k]), int(it1[k])) for k in bad[:8]]), "compactions", gg.compactions())
    gg.close()
