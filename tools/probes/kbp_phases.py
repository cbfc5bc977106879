"""Phase clocks of the persistent batched sweeps (kbp): shader clocks of thread 0 of every workgroup, per lock-step iteration."""
import sys, os, ctypes as C, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from miosqp_amd import qp, problems, _lib
import test_gpu_parity as T
width = 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
m = pr["A"].shape[0]
g = qp.OSQP()
g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=width, batch_pers=1))
g.set_integer_rows(pr["i_idx"], m)
g.set_root(l, u, 1e-3, 1e-3)
leaves = T._frontier(g, pr, l, u, width)[:width]
L = np.stack([lf.l for lf in leaves]); U = np.stack([lf.u for lf in leaves])
X = np.stack([lf.x for lf in leaves]); Y = np.stack([lf.y for lf in leaves])
g.solve_batch(L, U, X, Y)
lib = _lib.load()
G = 256
names = ["fwd sweep", "fwd reduce+store", "barrier 1", "x sweep", "x reduce+epilogue", "constraint tiles", "barrier 2"]
out = np.zeros(16 * G, dtype=np.uint64); nb = C.c_int32()
rc = lib.miosqp_qp_debug_timeline(g._h, 5, out.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * G, C.byref(nb))
o = out.reshape(G, 16).astype(np.float64)
it = o[:, 7]
print("rc", rc, "iterations", it[0], _lib.last_error() if rc else "")
tot = 0.0
for k, nm in enumerate(names):
    v = o[:, k] / it
    tot += np.median(v)
    print("  %-20s med %7.0f  min %7.0f  max %7.0f clocks / iteration   by member octile: %s" % (
        nm, np.median(v), v.min(), v.max(), " ".join("%6.0f" % np.median(v[(np.arange(G) >> 3) // 4 == x]) for x in range(8))))
print("  sum of medians %.0f clocks (2.4 GHz: %.2f us)" % (tot, tot / 2400.0))
