import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from miosqp_amd import problems, qp
from test_gpu_parity import _frontier

pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
m = pr["A"].shape[0]
def eng(cap, **kw):
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=cap, **kw))
    g.set_integer_rows(pr["i_idx"], m)
    g.set_root(l, u, 1e-3, 1e-3)
    return g
g = eng(256)
leaves = _frontier(g, pr, l, u, 512)[:512]
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
os.environ["MIOSQP_COMPACT"] = "0"
# one chunk only: iterates and residuals after 25 iterations, by batch size
ref = None
for cnt in (256, 320, 384, 448, 512):
    gg = eng(1024, max_iter=25, check_termination=25)
    rb = gg.solve_batch(L[:cnt], U[:cnt], X[:cnt], Y[:cnt])
    pri = np.array([i.pri_res for i in rb.infos]); dua = np.array([i.dua_res for i in rb.infos])
    if ref is None:
        ref = (rb.x.copy(), rb.y.copy(), pri.copy(), dua.copy())
    k = min(cnt, 256)
    print("count %d: max|dx| %.3e max|dy| %.3e  max|dpri| %.3e max|ddua| %.3e  status %s" % (
        cnt, np.abs(rb.x[:k] - ref[0][:k]).max(), np.abs(rb.y[:k] - ref[1][:k]).max(),
        np.abs(pri[:k] - ref[2][:k]).max(), np.abs(dua[:k] - ref[3][:k]).max(), np.unique(rb.status_val, return_counts=True)))
    gg.close()
s = [g.solve_node(L[k], U[k], X[k], Y[k]) for k in range(448)]
it1 = np.array([r.iter for r in s])
for kw, env in ((dict(fold=0), {}), (dict(), {"MIOSQP_BD_CFG": "44"})):
    os.environ.update(env)
    gg = eng(1024, **kw)
    rb = gg.solve_batch(L[:448], U[:448], X[:448], Y[:448])
    bad = [k for k in range(448) if rb.iter[k] != it1[k]]
    print(kw, env, "mismatches", len(bad), [(k, int(rb.iter[k]), int(it1[k])) for k in bad[:5]])
    gg.close()
