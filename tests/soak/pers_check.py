"""Development check of the persistent streaming solver (k_pers) against the CPU oracle, both factor forms, a few
sizes; prints microseconds per iteration of the single launch (no tests) next to the multi-kernel form.
usage: python tools/probes/pers_check.py [quick|cfg2|cfg5]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp
from oracle import oracle

def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))

PERS = int(os.environ.get("PERS_MODE", "1"))


def one(n, m, p, dens, seed, fold, check_oracle=True, reps=300):
    pr = problems.random_miqp(n, m, p, density=dens, seed=seed)
    A, l, u = problems.extended(pr)
    M = A.shape[0]
    t0 = time.time()
    g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=0, coop=0, pers=PERS, **problems.QP_SETTINGS)
    fs = g.factor_stats()
    print("n=%d m=%d p=%d dens=%g fold=%d: pers=%s setup %.2fs" % (n, m, p, dens, fold, fs["pers"], time.time() - t0), flush=True)
    if not fs["pers"]:
        return
    rng = np.random.RandomState(seed)
    x0, y0 = rng.randn(n), rng.randn(M)
    if check_oracle:
        o = oracle.OSQP(); o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        for k in (1, 2, 10, 50):
            g.warm_start(x=x0, y=y0); o.warm_start(x=x0, y=y0)
            xg, zg, yg = g.debug_iterate(k); o.iterate(k); xo, zo, yo = o.iterates()
            print("  k=%3d  rel x %.2e z %.2e y %.2e" % (k, rel(xg, xo), rel(zg, zo), rel(yg, yo)), flush=True)
        g.warm_start(x=np.zeros(n), y=np.zeros(M)); o.warm_start(x=np.zeros(n), y=np.zeros(M))
        rg, ro = g.solve(), o.solve()
        print("  solve: status %d/%d iter %d/%d rel x %.2e y %.2e" % (rg.info.status_val, ro.info.status_val, rg.info.iter,
              ro.info.iter, rel(rg.x, ro.x), rel(rg.y, ro.y)), flush=True)
    else:
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        rg = g.solve()
        print("  solve: status %d iter %d  device %.3f ms" % (rg.info.status_val, rg.info.iter, 1e3 * rg.info.device_time), flush=True)
    us, by = g.time_kernel(4, reps)
    print("  persistent launch: %.2f us / iteration (%.2f TB/s by the algorithmic bytes %d)" % (us, fs["bytes_per_iter"] / us * 1e-6, fs["bytes_per_iter"]), flush=True)
    g.close()
    g2 = qp.OSQP(); g2.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=0, coop=0, pers=0, **problems.QP_SETTINGS)
    us2, _ = g2.time_kernel(4, reps)
    print("  multi-kernel form : %.2f us / iteration" % us2, flush=True)
    for name, eng in (("persistent", None), ("multi-kernel", g2)):
        if eng is None:
            eng = qp.OSQP(); eng.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=0, coop=0, pers=PERS, **problems.QP_SETTINGS)
        best = None
        for rep in range(3):
            eng.warm_start(x=np.zeros(n), y=np.zeros(M))
            r = eng.solve()
            best = r.info.device_time if best is None else min(best, r.info.device_time)
        print("  whole solve, %-12s: %d iterations, %.3f ms on the device = %.2f us / iteration (tests included)" %
              (name, r.info.iter, 1e3 * best, 1e6 * best / r.info.iter), flush=True)
    g2.close()

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
if mode == "quick":
    for fold in (1, 0):
        one(60, 120, 30, 0.7, 11, fold)
        one(130, 260, 65, 0.7, 3, fold)
    one(500, 1000, 250, 0.7, 0, 1)
    one(500, 1000, 250, 0.7, 0, 0)
elif mode == "cfg2":
    one(500, 1000, 250, 0.7, 0, 1, reps=1000)
elif mode == "big":
    one(1000, 2000, 500, 0.7, 0, 1, check_oracle=False)
    one(1500, 3000, 750, 0.5, 0, 1, check_oracle=False)
elif mode == "cfg5x":
    one(8000, 16000, 4000, 0.01, 0, 0, check_oracle=False, reps=50)
elif mode == "cfg5":
    one(5000, 10000, 2500, 0.01, 0, 0, check_oracle=False, reps=100)
