"""Persistent batched sweeps (kbp1 / kbp) against the launches over random shapes inside kbp's limits (n <= 512, rows in the
products <= 1024): 256- and 320-leaf waves of random 0/1 fixings, bitwise comparison of solve_batch's outputs.
    python tools/probes/soak_kbp.py [shapes] [seed]"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from miosqp_amd import qp, problems

shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(shapes):
    n = int(rng.choice([40, 100, 200, 256, 300, 401, 500, 512]))
    m = int(rng.randint(max(8, n // 4), min(1024, 3 * n) + 1))
    p = int(rng.randint(2, max(3, n // 2)))
    dens = float(rng.choice([0.15, 0.4, 0.7]))
    noid = bool(rng.rand() < 0.25) and m + p <= 1024 and m + p + n <= 1536
    if noid:
        os.environ["MIOSQP_NO_IDROWS"] = "1"
    else:
        os.environ.pop("MIOSQP_NO_IDROWS", None)
    pr = problems.random_miqp(n, m, p, density=dens, seed=1000 + k)
    A, l, u = problems.extended(pr)
    B = int(rng.choice([256, 320, 192]))
    L = np.tile(l, (B, 1)); U = np.tile(u, (B, 1))
    for b in range(B):  # random 0/1 fixings of a few integer variables
        idx = rng.choice(p, size=min(p, 1 + b % 6), replace=False)
        val = rng.randint(0, 2, size=len(idx)).astype(float)
        L[b, m + idx] = val
        U[b, m + idx] = val
    X = np.zeros((B, n)); Y = np.zeros((B, m + p))
    out = []
    for bp in (0, 1):
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=max(B, 256), batch_pers=bp, max_iter=1000))
        g.set_integer_rows(pr["i_idx"], m)
        g.set_root(l, u, 1e-3, 1e-3)
        r = g.solve_batch(L, U, X, Y)
        out.append((r, g.factor_stats()["batch_pers"], g.batch_pers_fallbacks()))
        g.close()
    (a, _, _), (b_, on, fb) = out
    same = (np.array_equal(a.status_val, b_.status_val) and np.array_equal(a.iter, b_.iter) and
            np.array_equal(a.x, b_.x, equal_nan=True) and np.array_equal(a.y, b_.y, equal_nan=True))
    bad += 0 if same else 1
    print("n=%3d m=%4d p=%3d dens %.2f B=%d idrows=%s persistent=%s fallbacks=%d  statuses %s  max iter %d  -> %s" % (
        n, m, p, dens, B, not noid, on, fb, dict(zip(*np.unique(a.status_val, return_counts=True))), a.iter.max(),
        "same bits" if same else "DIFFERENT"), flush=True)
print("shapes with differences:", bad)
