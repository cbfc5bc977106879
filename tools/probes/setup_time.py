import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from miosqp_amd import qp, problems
for cfg in ("cfg2","cfg5"):
    pr = problems.random_miqp(**problems.CONFIGS[cfg], seed=0)
    A,l,u = problems.extended(pr)
    t=time.time(); g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u, **problems.QP_SETTINGS); print(cfg,'setup total %.3f s'%(time.time()-t))
