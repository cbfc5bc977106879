"""Phase clocks of the persistent streaming solver (k_pers): shader clocks of thread 0 of every workgroup, per iteration.
usage: python tools/probes/pers_phases.py n m p density fold [pers]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import qp, problems, _lib
a = sys.argv[1:]
n, m, p = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (500, 1000, 250)
dens = float(a[3]) if len(a) > 3 else 0.7
fold = int(a[4]) if len(a) > 4 else 1
pers = int(a[5]) if len(a) > 5 else 1
pr = problems.random_miqp(n, m, p, density=dens, seed=0)
A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, fold=fold, coop=0, resident=0, pers=pers, **problems.QP_SETTINGS)
assert g.factor_stats()["pers"]
g.warm_start(x=np.zeros(n), y=np.zeros(A.shape[0]))
lib = _lib.load()
G = 256
names = ["sparse fwd", "fwd wait", "fwd rows", "fwd epilogue", "bwd wait", "bwd rows", "bwd epilogue(+sparse)", "tests"]
for rep in range(1):
    out = np.zeros(16 * G, dtype=np.uint64); nb = C.c_int32()
    rc = lib.miosqp_qp_debug_timeline(g._h, 4, out.ctypes.data_as(C.POINTER(C.c_uint64)), 8 * G, C.byref(nb))
    o = out.reshape(G, 16).astype(np.float64)
    live = o[:, 8] > 0
    it = o[live, 8]
    print("rc", rc, "workgroups", int(live.sum()), "iterations", it[0])
    tot = 0.0
    for k, nm in enumerate(names):
        v = o[live, k] / it
        tot += np.median(v)
        print("  %-24s med %7.0f  min %7.0f  max %7.0f clocks / iteration" % (nm, np.median(v), v.min(), v.max()))
    print("  sum of medians %.0f clocks" % tot)
us, by = g.time_kernel(4, 1000)
print("time_kernel: %.2f us / iteration" % us)
# per-XCD view of the two row phases (workgroup b runs on XCD b % 8)
for k in (2, 5):
    v = o[live, k] / it
    idx = np.arange(G)[live]
    print("  %-10s by XCD:" % names[k], " ".join("%6.0f" % np.median(v[idx % 8 == x]) for x in range(8)))
    print("  %-10s by workgroup octile:" % names[k], " ".join("%6.0f" % np.median(v[(idx // 32) == x]) for x in range(8)))
