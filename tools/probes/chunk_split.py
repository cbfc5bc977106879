"""Streaming batch at config 2, 256 columns: wall time per chunk for several check_termination values -> the cost of
one lock-step iteration inside a chunk (slope) and of the chunk's serial part (intercept: test, harvest, refill)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems, stream  # noqa: E402

pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
pts = []
for chk in (25, 50, 100):
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9),
                dict(problems.QP_SETTINGS, max_batch=256, check_termination=chk, max_iter=4000))
    srch = stream.NativeStreamSearch(model, columns=256)
    rng = np.random.RandomState(12345)

    def steps(k):
        for _ in range(k):
            if srch.step() == 0:
                model.update_vectors(q=rng.randn(500), l=-2 + rng.rand(1000), u=2 + rng.rand(1000))
                srch.begin_instance()

    steps(15000 // chk)
    c0, n0, t0 = srch.chunks, srch.nodes, time.perf_counter()
    steps(10000 // chk)
    dt = time.perf_counter() - t0
    us = 1e6 * dt / (srch.chunks - c0)
    pts.append((chk, us))
    print("check_termination %3d: %.1f us per chunk, %.2f us per lock-step iteration, %d nodes" %
          (chk, us, us / chk, srch.nodes - n0))
    model.work.solver.close()
(a, ua), (b, ub) = pts[0], pts[-1]
slope = (ub - ua) / (b - a)
print("slope %.2f us per iteration, intercept %.1f us per chunk" % (slope, ua - slope * a))
