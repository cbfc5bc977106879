"""Times the batched sweep kernels at full width (config 2, 256 or N columns): python kbm_time.py [cols]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP()
g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=cols))
g.set_integer_rows(pr["i_idx"], 1000)
B = cols
rb = g.solve_batch(np.stack([l] * B), np.stack([u] * B), np.zeros((B, 500)), np.zeros((B, A.shape[0])))
f, fb = g.time_kernel(10, 200)
b, bb = g.time_kernel(11, 200)
print("cols %d ablate %s: kbm_fwd %.2f us, kbm_bwd %.2f us, sum %.2f us  (iters %d)" % (
    cols, os.environ.get("MIOSQP_BM_ABLATE", "0"), f, b, f + b, rb.iter[0]))
