import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from miosqp_amd import qp, problems
from golden_cases import load_power_converter
def timeit(g, n, M):
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    g.debug_iterate(10)
    out=[]
    for k in (200, 2200):
        t=time.perf_counter(); g.debug_iterate(k); out.append(time.perf_counter()-t)
    return (out[1]-out[0])/2000*1e6, out[0]*1e6
pr = problems.random_miqp(50,100,10,seed=0); A,l,u = problems.extended(pr)
for res in (0,1):
    g=qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u,resident=res, **problems.QP_SETTINGS)
    print('cfg1 resident',res, 'us/iter %.3f (200-iter call %.1f us)'%timeit(g,50,A.shape[0]), g.factor_stats())
pc=load_power_converter()
from miosqp_amd import bnb
A2,l2,u2 = bnb.add_bounds(pc['i_idx'], pc['i_l'], pc['i_u'], pc['A'], pc['l'], pc['u'][0])
for res in (0,1):
    g=qp.OSQP(); g.setup(pc['P'],pc['q'][0],A2,l2,u2,resident=res, **pc['qp_settings'])
    print('cfg4 resident',res, 'us/iter %.3f (200-iter call %.1f us)'%timeit(g,18,A2.shape[0]))
import ctypes as C
from miosqp_amd import _lib
lib=_lib.load(); lib.miosqp_qp_debug_clock.argtypes=[C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
g=qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u,resident=1, **problems.QP_SETTINGS)
g.warm_start(x=np.zeros(50), y=np.zeros(A.shape[0]))
for k in (100, 2000, 2000):
    g.debug_iterate(k)
    cy,tk=C.c_double(),C.c_double(); lib.miosqp_qp_debug_clock(g._h, C.byref(cy), C.byref(tk))
    print('iters',k,'cycles',cy.value,'wall_us',tk.value/100,'=> clock MHz', cy.value/(tk.value/100) if tk.value else 0, 'cycles/iter', cy.value/k)
