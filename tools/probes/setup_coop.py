import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import qp, problems
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0); A, l, u = problems.extended(pr)
for rep in range(3):
    for coop in (0, 1):
        g = qp.OSQP(); t = time.perf_counter(); g.setup(pr['P'], pr['q'], A, l, u, coop=coop, **problems.QP_SETTINGS)
        print('rep', rep, 'coop', coop, 'setup %.1f ms' % (1e3 * (time.perf_counter() - t)), flush=True)
