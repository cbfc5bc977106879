import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo')
from miosqp_amd import qp, problems, _lib
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
A,l,u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u, **problems.QP_SETTINGS)
lib=_lib.load()
lib.miosqp_qp_debug_timeline.argtypes=[C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32)]
for which,nb in ((0,500),(1,438)):
    for rep in range(3):
        out=np.zeros(2*nb,dtype=np.uint64); n=C.c_int32()
        rc=lib.miosqp_qp_debug_timeline(g._h, which, out.ctypes.data_as(C.POINTER(C.c_uint64)), nb, C.byref(n))
        st=out[0::2].astype(np.int64); en=out[1::2].astype(np.int64)
        t0=st.min()
        dur=(en-st)*10  # ns
        print('kernel',which,'rc',rc,'span_ns',(en.max()-t0)*10,'first_end',(en.min()-t0)*10,'last_start',(st.max()-t0)*10,
              'blockdur ns: min %d med %d max %d'%(dur.min(),np.median(dur),dur.max()))
        if rep==2:
            order=np.argsort(st)
            print(' start offsets (ns) deciles:', [(int(st[order[int(q*(nb-1))]]-t0)*10) for q in (0,.1,.25,.5,.75,.9,1)])
            # x rows vs c rows for bwd
            if which==1:
                print(' x-blocks dur med', np.median(dur[:125]), 'c-blocks dur med', np.median(dur[125:]))
