"""Soak of the symmetric-tile form of the persistent streaming solver (k_pers<false, true>): random shapes, densities, bounds
and warm starts; every instance on the tiles, on the row form (MIOSQP_PERS_SYM=0) and -- where the CPU can afford it -- on
the CPU restatement.  Status and iteration count must agree exactly, x and y to 1e-9 between the two device forms and to
the solution tolerance against the restatement; repeated solves of one engine must agree bit for bit.
usage: python tests/soak/soak_sym_tiles.py [instances] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp
from oracle import oracle

count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


bad = nodes = 0
edges, statuses = set(), {}
t0 = time.time()
for trial in range(count):
    big = trial % 6 == 5
    n = 2 * int(rng.randint(1000, 2900) if big else rng.randint(20, 600))
    m = int(rng.randint(50, 900))
    p = int(rng.randint(0, min(n, 200)))
    dens = float(rng.choice([0.004, 0.01, 0.03]) if big else rng.choice([0.05, 0.2, 0.7]))
    pr = problems.random_miqp(n, m, p, density=dens, seed=9000 + trial)
    A, l, u = problems.extended(pr)
    M = A.shape[0]
    l, u = l.copy(), u.copy()
    for j in rng.choice(M, size=min(M, 6), replace=False):  # some rows free, one-sided, tight or infeasible
        mode = int(rng.randint(0, 5))
        if mode == 0: l[j], u[j] = -1e30, 1e30
        elif mode == 1: u[j] = 1e30
        elif mode == 2: l[j] = u[j] = 0.5 * (l[j] + u[j])
        elif mode == 3: l[j] = -1e30
    x0, y0 = 0.3 * rng.randn(n), 0.3 * rng.randn(M)
    res = []
    for sym in ("1", "0"):
        os.environ["MIOSQP_PERS_SYM"] = sym
        g = qp.OSQP()
        g.setup(pr["P"], pr["q"], A, l, u, fold=0, resident=0, coop=0, pers=2, **problems.QP_SETTINGS)
        assert g.factor_stats()["tail_inverse"]
        tiles = g.tail_inverse_tiles()
        assert (tiles > 0) == (sym == "1"), (n, sym, tiles)
        g.warm_start(x=x0, y=y0)
        r1 = g.solve()
        g.warm_start(x=x0, y=y0)
        r2 = g.solve()
        if not (np.array_equal(r1.x, r2.x) and np.array_equal(r1.y, r2.y) and r1.info.iter == r2.info.iter):
            bad += 1
            print("NOT REPRODUCIBLE", n, m, p, dens, sym, flush=True)
        res.append(r1)
        g.close()
    a, b = res
    nodes += 1
    statuses[a.info.status_val] = statuses.get(a.info.status_val, 0) + 1
    ok = (a.info.status_val, a.info.iter) == (b.info.status_val, b.info.iter)
    if ok and a.info.status_val == 1:
        ok = rel(a.x, b.x) <= 1e-9 and rel(a.y, b.y) <= 1e-9
    if n <= 700:
        o = oracle.OSQP()
        o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
        o.warm_start(x=x0, y=y0)
        ro = o.solve()
        ok = ok and (a.info.status_val, a.info.iter) == (ro.info.status_val, ro.info.iter)
        if ok and ro.info.status_val == 1:
            ok = rel(a.x, ro.x) <= 1e-6 and rel(a.y, ro.y) <= 1e-6
    if not ok:
        bad += 1
        print("MISMATCH n=%d m=%d p=%d dens=%g: tiles %d/%d, rows %d/%d" % (n, m, p, dens, a.info.status_val, a.info.iter,
                                                                         b.info.status_val, b.info.iter), flush=True)
print("symmetric tiles against the row form (and the CPU restatement for n <= 700): %d instances, n from 40 to 5800, statuses %s, "
      "%d mismatches, %.0f s" % (nodes, statuses, bad, time.time() - t0))
sys.exit(1 if bad else 0)
