"""Per-workgroup phase stamps of the batched forward sweep (kbm_fwd): python kbm_timeline.py [cols]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp, _lib
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP()
g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, max_batch=cols, max_iter=50))
g.set_integer_rows(pr["i_idx"], 1000)
B = cols
g.solve_batch(np.stack([l] * B), np.stack([u] * B), np.zeros((B, 500)), np.zeros((B, A.shape[0])))
lib = _lib.load()
nwg = 32 * (cols // 32)
for rep in range(1):
    out = np.zeros(8 * nwg, dtype=np.uint64); n = C.c_int32()
    rc = lib.miosqp_qp_debug_timeline(g._h, 3, out.ctypes.data_as(C.POINTER(C.c_uint64)), 4 * nwg, C.byref(n))
    t = out.reshape(nwg, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    ns = lambda a: np.round(np.percentile(a, [0, 10, 50, 90, 100]) * 10).astype(int)
    print("rc", rc, "workgroups", len(t), "span ns", (t[:, 5].max() - t0) * 10)
    print(" start offset        ", ns(t[:, 0] - t0))
    print(" prologue (issue)    ", ns(t[:, 1] - t[:, 0]))
    print(" first data arrives  ", ns(t[:, 2] - t[:, 1]))
    print(" sweep after that    ", ns(t[:, 3] - t[:, 2]))
    print(" reduce              ", ns(t[:, 4] - t[:, 3]))
    print(" store               ", ns(t[:, 5] - t[:, 4]))
    print(" total per workgroup ", ns(t[:, 5] - t[:, 0]))
