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
    out=np.zeros(8*T,dtype=np.uint64); n=C.c_int32()
    rc=lib.miosqp_qp_debug_timeline(g._h, 2, out.ctypes.data_as(C.POINTER(C.c_uint64)), 4*T, C.byref(n))
    o=out.reshape(T,8).astype(np.float64)
    it=o[:,3]; nc=np.maximum(o[:,6],1)
    print('rc',rc,'iters',it[0],'tests',o[0,6])
    print(' per-iteration clocks (thread 0): reduce med %.0f | update+publish med %.0f | gather med %.0f' % tuple(np.median(o[:,k]/it) for k in range(3)))
    print('   max: reduce %.0f (wg %d) | update+publish %.0f (wg %d) | gather min %.0f (wg %d)' % ((o[:,0]/it).max(), (o[:,0]/it).argmax(), (o[:,1]/it).max(), (o[:,1]/it).argmax(), (o[:,2]/it).min(), (o[:,2]/it).argmin()))
    print(' per-test clocks: operands+rows med %.0f max %.0f | norms exchange med %.0f max %.0f' % (np.median(o[:,4]/nc),(o[:,4]/nc).max(),np.median(o[:,5]/nc),(o[:,5]/nc).max()))
