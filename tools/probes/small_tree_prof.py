"""MIOSQP_TREE_PROF=1 breakdown of the one-workgroup tree kernel on the reference grid's shapes (one instance each)."""
import os, sys, time
os.environ["MIOSQP_TREE_PROF"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems
np.random.seed(0)
for (n, m, p) in [(10, 100, 2), (50, 25, 5), (100, 50, 2)]:
    pr = problems.random_miqp(n, m, p, density=0.7, reseed=False)
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    root = model.work.leaves[0]
    for k in range(2):
        r = model.work.solver.solve_tree(root.l, root.u, root.x, root.y, np.inf, None, 1, 1000)
        if r is None:
            print("n=%d m=%d: not covered" % (n, m)); break
        sys.stderr.flush()
        print("n=%d m=%d p=%d call %d: device %.3f ms, %d nodes, %d iterations, stats %s" % (
            n, m, p, k, 1e3 * r.info.device_time, r.info.nodes, r.info.osqp_iter, model.work.solver.factor_stats()), flush=True)
