"""Set-up wall time at config 2: rho fixed against rho chosen at set-up, medians over alternating runs.
usage: setup_ratio.py [runs]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from miosqp_amd import problems, qp
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 9
t = {0.1: [], "auto": []}
for k in range(runs + 1):
    for rho in (0.1, "auto"):
        t0 = time.perf_counter()
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, rho=rho))
        g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
        dt = time.perf_counter() - t0
        g.close()
        if k:
            t[rho].append(dt)
a, b = np.median(t[0.1]), np.median(t["auto"])
print("set-up, median of %d: rho fixed %.1f ms (min %.1f), rho chosen at set-up %.1f ms (min %.1f): x %.3f (of the minima: x %.3f)"
      % (runs, 1e3 * a, 1e3 * min(t[0.1]), 1e3 * b, 1e3 * min(t["auto"]), b / a, min(t["auto"]) / min(t[0.1])))
