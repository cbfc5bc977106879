"""Soak of single relaxations over the engine forms: random shapes (odd and even sizes, n+M from 10 to 900), the HIP
engine's automatic form against the CPU oracle from a random warm start and on a branched child: status, iteration
count, x, y.  usage: soak_nodes.py [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp  # noqa: E402
from oracle import oracle  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(77)
bad, forms, t0 = 0, {}, time.time()


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


for k in range(count):
    small = os.environ.get("SOAK_SMALL")  # SOAK_SMALL=1: shapes of the one-workgroup kernels (n+M <= 192)
    n = int(rng.randint(3, 70 if small else 260))
    m = int(rng.randint(1, 100 if small else 520))
    p = int(rng.randint(1, min(n, 20 if small else 40) + 1))
    pr = problems.random_miqp(n, m, p, seed=5000 + k, density=float(rng.choice([0.05, 0.3, 0.7])))
    A, l, u = problems.extended(pr)
    g, o = qp.OSQP(), oracle.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    fs = g.factor_stats()
    key = "resident" if fs["resident"] else "coop" if fs["coop"] else "fold" if fs["fold"] else "factor"
    forms[key] = forms.get(key, 0) + 1
    x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(A.shape[0])
    lo, hi = l.copy(), u.copy()
    for step in range(2):
        g.update(l=lo, u=hi); o.update(l=lo, u=hi)
        g.warm_start(x=x0, y=y0); o.warm_start(x=x0, y=y0)
        rg, ro = g.solve(), o.solve()
        ok = (rg.info.status_val, rg.info.iter) == (ro.info.status_val, ro.info.iter)
        if ok and ro.info.status_val == 1:
            ok = rel(rg.x, ro.x) <= 1e-7 and rel(rg.y, ro.y) <= 1e-7
        if not ok:
            bad += 1
            print("MISMATCH case %d (n %d m %d p %d, %s) step %d: status %d/%d iter %d/%d" %
                  (k, n, m, p, key, step, rg.info.status_val, ro.info.status_val, rg.info.iter, ro.info.iter))
        j = m + int(rng.randint(p))  # fix one integer variable: a child node
        v = float(rng.randint(2))
        lo, hi = lo.copy(), hi.copy()
        lo[j] = hi[j] = v
        if ro.info.status_val == 1:
            x0, y0 = ro.x, ro.y
    g.close()
print("%d cases x 2 solves, %d mismatches, %.1f s, forms %s" % (count, bad, time.time() - t0, forms))
