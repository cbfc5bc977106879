"""Fixed cost of one k_coop launch: event time of launches of 1, 2, 10, 100, 1000 iterations."""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import qp, problems
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0); A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, **problems.QP_SETTINGS)
g.warm_start(x=np.zeros(A.shape[1]), y=np.zeros(A.shape[0]))
for reps in (1, 2, 10, 100, 1000, 1, 10):
    us, _ = g.time_kernel(4, reps)
    print('launch of %4d iterations: %.1f us total' % (reps, us * reps))
