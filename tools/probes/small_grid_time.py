"""Where a tiny MIQP's wall time goes on the HIP path: setup (engine creation) vs solve (one tree launch)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems

def run(n, m, p, reps=20):
    np.random.seed(0)
    ts, tv, tr = [], [], []
    for _ in range(reps):
        pr = problems.random_miqp(n, m, p, density=0.7, reseed=False)
        t0 = time.perf_counter()
        model = bnb.MIOSQP()
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                    dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        t1 = time.perf_counter()
        res = model.solve()
        t2 = time.perf_counter()
        ts.append(1e3 * (t1 - t0)); tv.append(1e3 * (t2 - t1)); tr.append(1e3 * res.run_time)
    print("n=%d m=%d p=%d setup %.3f ms (work.setup_time %.3f) solve %.3f ms run_time %.3f ms osqp %.3f" % (
        n, m, p, np.median(ts), 1e3 * model.work.setup_time, np.median(tv), np.median(tr), 1e3 * res.osqp_solve_time))

run(10, 5, 2); run(10, 5, 2); run(10, 100, 2); run(50, 25, 5)
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = problems.random_miqp(10, 5, 2, density=0.7, seed=3)
    def go():
        for _ in range(200):
            model = bnb.MIOSQP()
            model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                        dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
            model.solve()
    cProfile.run("go()", "/tmp/prof.out")
    pstats.Stats("/tmp/prof.out").sort_stats("cumulative").print_stats(35)
