"""k_coop at config 2: launch duration (HIP events around the launch) for solves of fixed length (eps tiny, max_iter =
25 ... 800) -> per-iteration slope and per-launch intercept (start-up + exit; the speculative iterations behind a
converged test are not in it: these solves end at max_iter)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import problems, qp  # noqa: E402

pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
pts = []
for mi in (25, 50, 100, 200, 400, 800):
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, eps_abs=1e-14, eps_rel=1e-14, max_iter=mi))
    n, M = A.shape[1], A.shape[0]
    for _ in range(3):
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        g.solve()
    g.loop_stats(reset=True)
    reps = 20
    for _ in range(reps):
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        r = g.solve()
    ms, it = g.loop_stats()
    pts.append((mi, 1e3 * ms / reps))
    print("max_iter %4d: %.1f us per launch (%d iterations)" % (mi, 1e3 * ms / reps, r.info.iter))
    g.close()
(a, ua), (b, ub) = pts[1], pts[-1]
s = (ub - ua) / (b - a)
print("slope %.3f us per iteration, intercept %.1f us per launch" % (s, ua - s * a))
