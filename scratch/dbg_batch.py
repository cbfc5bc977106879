import sys, numpy as np
sys.path.insert(0,'/root/repo')
from miosqp_amd import qp, problems
pr = problems.random_miqp(50,100,25,seed=2)
A,l,u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u,max_batch=64, **problems.QP_SETTINGS)
g.set_integer_rows(pr['i_idx'], 100)
n,M=50,A.shape[0]
L=np.stack([l]*5); U=np.stack([u]*5); U[1,-1]=0; U[2,-2]=0; L[3,-3]=1
X=np.zeros((5,n)); Y=np.zeros((5,M))
r = g.solve_batch(L,U,X,Y)
print(r.status_val, r.iter, r.lower)
for k in range(5):
    r1=g.solve_node(L[k],U[k],X[k],Y[k]); print(k, r1.status_val, r1.iter, r1.lower, np.abs(r1.x-r.x[k]).max())
print(g.batch_stats())
print(g.time_kernel(10,10))
