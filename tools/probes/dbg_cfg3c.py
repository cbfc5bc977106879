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
leaves = _frontier(g, pr, l, u, 512)[:512]
print(len(leaves))
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
s = [g.solve_node(L[k], U[k], X[k], Y[k]) for k in range(len(leaves))]
it1 = np.array([r.iter for r in s]); st1 = np.array([r.status_val for r in s])
print("statuses", np.unique(st1, return_counts=True))
for cap, compact, cnt in ((1024, "1", 448), (1024, "0", 448), (512, "1", 448), (1024, "1", 256), (1024, "1", 64), (1024, "0", 64), (512, "0", 64), (320, "0", 64), (320, "1", 300)):
    os.environ["MIOSQP_COMPACT"] = compact
    gg = eng(cap)
    rb = gg.solve_batch(L[:cnt], U[:cnt], X[:cnt], Y[:cnt])
    bad = [k for k in range(cnt) if (rb.status_val[k], rb.iter[k]) != (st1[k], it1[k])]
    print("cap %d compact %s count %d: %d mismatches; first %s" % (cap, compact, cnt, len(bad),
          [(k, int(rb.iter[k]), int(it1[k])) for k in bad[:6]]), "compactions", gg.compactions())
    gg.close()
