import os, sys, time
os.environ["MIOSQP_SETUP_TIMING"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems
np.random.seed(0)
for k in range(4):
    pr = problems.random_miqp(10, 5, 2, density=0.7, reseed=False)
    sys.stderr.write("---- instance %d\n" % k)
    t0 = time.perf_counter()
    model = bnb.MIOSQP()
    model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"],
                dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS))
    t1 = time.perf_counter()
    res = model.solve()
    t2 = time.perf_counter()
    del model
    t3 = time.perf_counter()
    sys.stderr.write("python: setup %.3f solve %.3f del %.3f ms\n" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2)))
