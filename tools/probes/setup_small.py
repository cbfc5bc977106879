"""Setup / solve / cleanup wall time of small instances (reference grid sizes), warm process."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, bnb, qp
for (n, m, p) in ((10, 5, 2), (50, 25, 5), (100, 50, 2), (150, 300, 20)):
    pr = problems.random_miqp(n, m, p, seed=1)
    A, l, u = problems.extended(pr)
    ts, tn, tc, tm = [], [], [], []
    for rep in range(6):
        t0 = time.perf_counter()
        g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        g.set_integer_rows(pr["i_idx"], m); g.set_root(l, u, 1e-3, 1e-3)
        t1 = time.perf_counter()
        r = g.solve_node(l, u, np.zeros(n), np.zeros(A.shape[0]))
        t2 = time.perf_counter()
        g.close()
        t3 = time.perf_counter()
        model = bnb.MIOSQP()
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        res = model.solve()
        t4 = time.perf_counter()
        model.work.solver.close()
        ts.append(t1 - t0); tn.append(t2 - t1); tc.append(t3 - t2); tm.append(t4 - t3)
    f = lambda a: "%.3f" % (1e3 * np.median(a[1:]))
    print("n=%d m=%d p=%d: setup %s ms, first node %s ms (%d it), cleanup %s ms, whole MIOSQP setup+solve %s ms (%d nodes); stats %s" % (
        n, m, p, f(ts), f(tn), r.iter, f(tc), f(tm), model.work.iter_num - 1, {k: v for k, v in g.__dict__.items() if k in ()}))
