"""Config 5 (n = 5000, m = 10000, 1 % dense, factor form, tail as S^-1): us per iteration of k_pers, iterations only
(time_kernel) and over a whole node launch with its tests.  Environment switches of the library apply (MIOSQP_PERS_*).
usage: python tools/probes/cfg5_time.py [n m p density]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import qp, problems  # noqa: E402

a = sys.argv[1:]
n, m, p = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (5000, 10000, 2500)
dens = float(a[3]) if len(a) > 3 else 0.01
pr = problems.random_miqp(n, m, p, density=dens, seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP()
g.setup(pr["P"], pr["q"], A, l, u, fold=0, coop=0, resident=0, pers=2, **problems.QP_SETTINGS)
fs = g.factor_stats()
assert fs["pers"], fs
us = [g.time_kernel(4, 500)[0] for _ in range(3)]
g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
r = g.solve()
g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
t0 = time.perf_counter()
r = g.solve()
dt = time.perf_counter() - t0
print("PF=%s DEAL=%s: time_kernel %s us/it; solve: %d iterations, status %d, %.2f us/it end to end, device %.2f us/it"
      % (os.environ.get("MIOSQP_PERS_PF", "-"), os.environ.get("MIOSQP_PERS_DEAL", "-"), " ".join("%.2f" % v for v in us),
         r.info.iter, r.info.status_val, 1e6 * dt / r.info.iter, 1e3 * r.info.device_time / r.info.iter))
