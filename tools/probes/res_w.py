"""LDS-resident solver: the product-form loop against the register-resident explicit-inverse loop (MIOSQP_RES_W), same
process, alternating; long fixed-length solves so that the per-iteration cost shows (us per iteration from the launch's
own HIP events)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import problems, qp  # noqa: E402

st = dict(problems.QP_SETTINGS, eps_abs=1e-14, eps_rel=1e-14, max_iter=2000, resident=1, coop=0,
          check_termination=int(os.environ.get("CHK", "25")))
for (n, m, p) in [(10, 20, 5), (25, 50, 10), (50, 100, 10), (60, 110, 20)]:
    pr = problems.random_miqp(n, m, p, seed=0)
    A, l, u = problems.extended(pr)
    out = []
    for rep in range(2):
        for w in ("0", "1"):
            os.environ["MIOSQP_RES_W"] = w
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, **st)
            g.loop_stats(reset=True)
            for _ in range(5):
                g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
                r = g.solve()
            ms, it = g.loop_stats()
            out.append((w, 1e3 * ms / max(1, it), r.info.iter, float(r.info.obj_val)))
            g.close()
    print("N %3d:" % (n + A.shape[0]), "  ".join("W=%s %.3f us/it (%d it, obj %.9g)" % o for o in out))
