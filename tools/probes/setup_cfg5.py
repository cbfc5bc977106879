import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import problems, qp
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
pr = problems.random_miqp(**problems.CONFIGS[cfg], seed=0)
A, l, u = problems.extended(pr)
for rep in range(3):
    t0 = time.time()
    g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    print("setup %.3f s" % (time.time() - t0), flush=True)
    g.close()
