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
    return g
g = eng(256)
g.set_root(l, u, 1e-3, 1e-3)
leaves = _frontier(g, pr, l, u, 512)[:512]
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
os.environ["MIOSQP_COMPACT"] = "0"
for iters in (1, 2):
    out = {}
    for cnt in (384, 448):
        gg = eng(1024, max_iter=iters, check_termination=iters)
        rb = gg.solve_batch(L[:cnt], U[:cnt], X[:cnt], Y[:cnt])
        out[cnt] = (rb.x.copy(), rb.y.copy())
        gg.close()
    dx = np.abs(out[384][0] - out[448][0][:384]); dy = np.abs(out[384][1] - out[448][1][:384])
    print("iters", iters, "max dx %.3e dy %.3e" % (dx.max(), dy.max()))
    bc = np.where((dx.max(axis=1) > 1e-12) | (dy.max(axis=1) > 1e-12))[0]
    print(" bad columns:", len(bc), bc[:40])
    if len(bc):
        c = bc[0]
        print(" col", c, "bad x rows", np.where(dx[c] > 1e-12)[0][:40], "bad y rows", np.where(dy[c] > 1e-12)[0][:40], "n bad y", (dy[c] > 1e-12).sum())
gg = eng(1024, max_iter=1, check_termination=1)
a = gg.solve_batch(L[:384], U[:384], X[:384], Y[:384]); ya = a.y.copy()
b = gg.solve_batch(L[:448], U[:448], X[:448], Y[:448]); yb = b.y.copy()
dy = np.abs(ya - yb[:384])
bc = (dy.max(axis=1) > 1e-12)
print("bad cols mask by 16:", ["%d:%d" % (i, bc[i*16:(i+1)*16].sum()) for i in range(24)])
br = (dy > 1e-12).sum(axis=0)
print("bad count per 16-row block:", [int(br[i*16:(i+1)*16].sum()) for i in range(79)])
c = 0
rows = np.where(dy[c] > 1e-12)[0][:6]
for r in rows:
    print("col0 row", r, "y0 %.6f y384 %.6f y448 %.6f" % (Y[c][r], ya[c][r], yb[c][r]))
