import os, sys, time
os.environ["MIOSQP_SETUP_TIMING"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
from miosqp_amd import problems, bnb, qp
for (n, m, p) in ((10, 5, 2), (10, 100, 2)):
    pr = problems.random_miqp(n, m, p, seed=1)
    for rep in range(3):
        sys.stderr.write("---- n=%d m=%d rep %d\n" % (n, m, rep)); sys.stderr.flush()
        t0 = time.perf_counter()
        model = bnb.MIOSQP()
        model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
        t1 = time.perf_counter()
        res = model.solve()
        t2 = time.perf_counter()
        model.work.solver.close()
        t3 = time.perf_counter()
        sys.stderr.write("== setup %.3f ms solve %.3f ms close %.3f ms nodes %d\n" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), model.work.iter_num - 1)); sys.stderr.flush()
