"""Timeline of one termination test on the tester workgroups of k_coop (third test of a 1000-iteration launch):
100 MHz wall-clock stamps relative to the moment workgroup 0 published the test's operands."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import qp, problems, _lib
dims = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (500, 1000, 250)
pr = problems.random_miqp(*dims, seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
g.warm_start(x=np.zeros(A.shape[1]), y=np.zeros(A.shape[0]))
lib = _lib.load()
lib.miosqp_qp_debug_timeline.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32)]
T = (A.shape[0] + A.shape[1] + 7) // 8
NB = T + 32
out = np.zeros(8 * NB, dtype=np.uint64); n = C.c_int32()
rc = lib.miosqp_qp_debug_timeline(g._h, 2, out.ctypes.data_as(C.POINTER(C.c_uint64)), 4 * NB, C.byref(n))
o = out.reshape(NB, 8).astype(np.int64)
t0 = o[0, 5]
print('rc', rc, 'grid', T, 'iterations', o[0, 3])
print('workgroup 0: operands published at 0, decision in hand after %.2f us' % ((o[0, 7] - t0) / 100.0))
ts = o[T:]
ts = ts[ts[:, 0] > 0]
names = ['first operand seen', 'operands gathered', 'rows done (wave 0)', 'all waves done', 'decision known']
for k, nm in enumerate(names):
    v = (ts[:, k] - t0) / 100.0
    print(' testers: %-22s min %6.2f med %6.2f max %6.2f us' % (nm, v.min(), np.median(v), v.max()))
