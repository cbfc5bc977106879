"""The persistent stream kernel (kbs) at config 3: node-iterations/s of the compiled stream driver and the kernel's own phase
clocks (MIOSQP_KBS_PROF=1: shader clocks of thread 0 of every workgroup in iterations / test / harvest + refill).
usage: kbs_phases.py [rounds] [columns]"""
import os
import sys
import time
import ctypes as C
import numpy as np
os.environ.setdefault("MIOSQP_KBS_PROF", "1")
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from miosqp_amd import bnb, problems, stream, _lib

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
model = bnb.MIOSQP()
model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st,
            dict(problems.QP_SETTINGS, max_batch=cols))
eng = model.work.solver
ss = stream.NativeStreamSearch(model, columns=cols)
rng = np.random.RandomState(7)
m = pr["A"].shape[0]


def steps(count):
    for _ in range(count):
        if ss.step() == 0:
            model.update_vectors(q=rng.randn(len(pr["q"])), l=-2 + rng.rand(m), u=2 + rng.rand(m))
            ss.begin_instance()


steps(700)
lib = _lib.load()
out = np.zeros(32 * 256, dtype=np.uint64)
nb = C.c_int32()
have_prof = lib.miosqp_qp_debug_timeline(eng._h, 6, out.ctypes.data_as(C.POINTER(C.c_uint64)), 16 * 256, C.byref(nb)) == 0
eng.batch_stats(reset=True)
n0, i0 = ss.nodes, ss.iters
t0 = time.perf_counter()
steps(rounds)
ss.step(rounds=-1)  # the launches in flight, their digests absorbed
dt = time.perf_counter() - t0
ms, lock, nodeit = eng.batch_stats()
print("columns %d: %.3f M node-it/s end to end, %.1f k nodes/s, %.1f it/node; device %.2f us per lock-step iteration, occupancy %.3f, "
      "fallbacks %d, kbs %s" % (cols, (ss.iters - i0) / dt * 1e-6, (ss.nodes - n0) / dt * 1e-3, (ss.iters - i0) / max(1, ss.nodes - n0),
                                1e3 * ms / max(1, lock), nodeit / float(max(1, cols * lock)), eng.batch_pers_fallbacks(),
                                eng.factor_stats()["batch_pers"]))
if have_prof and lib.miosqp_qp_debug_timeline(eng._h, 6, out.ctypes.data_as(C.POINTER(C.c_uint64)), 16 * 256, C.byref(nb)) == 0:
    o = out[:8 * 256].reshape(256, 8).astype(np.float64)
    ip = out[8 * 256:16 * 256].reshape(256, 8).astype(np.float64)
    fp = out[16 * 256:].reshape(256, 16).astype(np.float64)
    ch = np.maximum(o[:, 4], 1.0)
    for k, nm in enumerate(["iterations", "test", "harvest + refill"]):
        v = o[:, k] / ch
        print("  %-18s med %8.0f  min %8.0f  max %8.0f clocks per chunk (%.1f us at 2.4 GHz)" % (nm, np.median(v), v.min(), v.max(),
                                                                                           np.median(v) / 2400.0))
    print("  chunks per workgroup: med %.0f min %.0f max %.0f" % (np.median(o[:, 4]), o[:, 4].min(), o[:, 4].max()))
    its = 25.0 * ch
    for k, nm in enumerate(["fwd sweep", "fwd reduce+store", "barrier 1", "x sweep", "x epilogue", "constraint tiles", "barrier 2"]):
        v = ip[:, k] / its
        print("    %-18s med %7.0f  min %7.0f  max %7.0f clocks per iteration" % (nm, np.median(v), v.min(), v.max()))
    names = ["test jobs", "row pieces + fold", "barrier", "members' fold + decide", "harvest rows + claim", "barrier", "table + harvest jobs + fold",
             "prepare", "barrier", "commit", "z jobs + stores", "barrier", "counters"]
    for k, nm in enumerate(names):
        v = fp[:, k] / ch
        print("    %-28s med %7.0f  min %7.0f  max %7.0f clocks per chunk; member 0: %7.0f" % (nm, np.median(v), v.min(), v.max(), np.median(v[:8])))
    mem = np.arange(256) >> 3
    for k in (2, 5, 8, 11):
        v = fp[:, k] / ch
        print("    barrier (phase %d) by member: %s" % (k, " ".join("%d" % (np.median(v[mem == mm]) / 100) for mm in range(32))))
    for k in (9, 10):
        v = fp[:, k] / ch
        print("    phase %d by member: %s" % (k, " ".join("%d" % (np.median(v[mem == mm]) / 100) for mm in range(32))))
