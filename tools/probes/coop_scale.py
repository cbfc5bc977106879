"""us/iter of the cooperative solver versus problem size (number of workgroups in the exchange)."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import qp, problems
for (n, m, p) in [(40, 80, 8), (75, 150, 30), (150, 300, 60), (300, 600, 120), (400, 800, 200), (500, 1000, 250), (580, 1160, 290)]:
    pr = problems.random_miqp(n, m, p, seed=0); A, l, u = problems.extended(pr)
    M = A.shape[0]
    res = []
    for coop in (1, 0):
        g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, coop=coop, resident=0, **problems.QP_SETTINGS)
        g.warm_start(x=np.zeros(n), y=np.zeros(M)); g.debug_iterate(10)
        out = []
        for k in (200, 2200):
            t = time.perf_counter(); g.debug_iterate(k); out.append(time.perf_counter() - t)
        res.append((out[1] - out[0]) / 2000 * 1e6)
    print('N %5d  T %3d  coop %.3f us/iter   fold %.3f us/iter' % (n + M, (n + M + 7) // 8, res[0], res[1]), flush=True)
