"""solve_tree on a fresh engine and on a warm one: Python wall, the C call's own wall (info.run_time), device time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems
np.random.seed(0)
for (n, m, p) in [(10, 5, 2), (10, 100, 2), (50, 25, 5), (50, 200, 10)]:
    for rep in range(3):
        pr = problems.random_miqp(n, m, p, density=0.7, reseed=False)
        model = bnb.MIOSQP()
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        root = model.work.leaves[0]
        out = []
        for k in range(3):
            t0 = time.perf_counter()
            r = model.work.solver.solve_tree(root.l, root.u, root.x, root.y, np.inf, None, 1, 1000)
            t1 = time.perf_counter()
            out.append("call %d: python %.3f  C %.3f  device %.3f ms (%d nodes)" % (k, 1e3 * (t1 - t0), 1e3 * r.info.run_time, 1e3 * r.info.device_time, r.info.nodes))
        print("n=%d m=%d p=%d  " % (n, m, p) + " | ".join(out))
