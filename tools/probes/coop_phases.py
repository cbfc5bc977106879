import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from miosqp_amd import qp, problems, _lib
dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (500, 1000, 250)
pr = problems.random_miqp(*dims, seed=0)
A,l,u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u, coop=1, resident=0, **problems.QP_SETTINGS)
g.warm_start(x=np.zeros(A.shape[1]), y=np.zeros(A.shape[0]))
lib=_lib.load()
lib.miosqp_qp_debug_timeline.argtypes=[C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32)]
T=(A.shape[0]+A.shape[1]+7)//8
for rep in range(2):
    out=np.zeros(4*T,dtype=np.uint64); n=C.c_int32()
    rc=lib.miosqp_qp_debug_timeline(g._h, 2, out.ctypes.data_as(C.POINTER(C.c_uint64)), 2*T, C.byref(n))
    o=out.reshape(T,4).astype(np.float64)/1000
    print('rc',rc,'per-iteration clocks (thread 0): reduce med %.0f max %.0f | update+publish med %.0f max %.0f | gather med %.0f min %.0f max %.0f | poll rounds med %.2f max %.2f'%(
        np.median(o[:,0]),o[:,0].max(),np.median(o[:,1]),o[:,1].max(),np.median(o[:,2]),o[:,2].min(),o[:,2].max(),np.median(o[:,3]),o[:,3].max()))
    print(' total clocks/iter med', np.median(o[:,:3].sum(1)))
