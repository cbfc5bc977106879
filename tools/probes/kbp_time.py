"""Device time per lock-step iteration of a 256-column wave, compaction off: the launches vs kbp; bitwise comparison."""
import os, sys, time
os.environ["MIOSQP_COMPACT"] = "0"
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from miosqp_amd import qp, problems
import test_gpu_parity as T
width = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
m = pr["A"].shape[0]
def engine(bp):
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
for tag, bp in (("launches", 0), ("kbp", 1)):
    g = g0 if bp == 0 else engine(1)
    g.solve_batch(L, U, X, Y)
    g.batch_stats(reset=True)
    out[bp] = g.solve_batch(L, U, X, Y)
    ms, bi, ni = g.batch_stats()
    print("%-10s %7.2f us per lock-step iteration incl. one test per 25 (%d iterations; persistent %s, fallbacks %d)" % (
        tag, 1e3 * ms / max(1, bi), bi, g.factor_stats()["batch_pers"], g.batch_pers_fallbacks()), flush=True)
a, b = out[0], out[1]
ok = a.status_val == 1
print("status", np.array_equal(a.status_val, b.status_val), "iter", np.array_equal(a.iter, b.iter),
      "x", np.array_equal(a.x[ok], b.x[ok]), "y", np.array_equal(a.y[ok], b.y[ok]), "lower", np.array_equal(a.lower[ok], b.lower[ok]))
