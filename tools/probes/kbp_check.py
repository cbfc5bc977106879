"""kbp (a chunk's lock-step iterations as one persistent launch) against the launches on the same wave: bitwise
comparison of solve_batch's outputs, and the device time per lock-step iteration of both.
    python tools/probes/kbp_check.py [width]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from miosqp_amd import qp, problems
import test_gpu_parity as T

width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
m = pr["A"].shape[0]

def engine(bp, compact=True):
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=width, batch_pers=bp))
    g.set_integer_rows(pr["i_idx"], m)
    g.set_root(l, u, 1e-3, 1e-3)
    return g

g0 = engine(0)
leaves = T._frontier(g0, pr, l, u, width)[:width]
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
out = {}
for bp in (0, 1):
    g = g0 if bp == 0 else engine(1)
    g.solve_batch(L, U, X, Y)  # warm-up (graph capture)
    g.batch_stats(reset=True)
    t0 = time.perf_counter()
    r = g.solve_batch(L, U, X, Y)
    dt = time.perf_counter() - t0
    st = g.batch_stats()
    print("batch_pers=%d: %d leaves, wall %.1f ms, batch stats %s, factor stats kbp=%s" % (
        bp, len(leaves), 1e3 * dt, st, g.factor_stats().get("batch_pers")), flush=True)
    out[bp] = r
a, b = out[0], out[1]
print("status equal", np.array_equal(a.status_val, b.status_val), " iter equal", np.array_equal(a.iter, b.iter))
ok = a.status_val == 1
print("x bitwise equal", np.array_equal(a.x[ok], b.x[ok]), " max |dx|", np.abs(a.x[ok] - b.x[ok]).max())
print("y bitwise equal", np.array_equal(a.y[ok], b.y[ok]), " max |dy|", np.abs(a.y[ok] - b.y[ok]).max())
print("lower equal", np.array_equal(a.lower[ok], b.lower[ok]))
