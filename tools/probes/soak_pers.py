"""Soak of the persistent streaming solver: (a) many node relaxations at config 2 in the lean product-form kernel, the
general product-form kernel and the factor form, each against the multi-kernel form of the same engine settings on the same
inputs (status and iteration count equal, x and y to 1e-9 relative); (b) a branch-and-bound search on each; (c) a few hundred
solves of random small shapes.  A hand-off that fails once in 10^5 rounds shows up here as a time-out or a mismatch.
usage: soak_pers.py [nodes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems, qp  # noqa: E402

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
A, l, u = problems.extended(pr)
n, M, m = 500, A.shape[0], 1000
bad = 0
for name, kw, env in (("lean product form", dict(fold=1, pers=1), {}), ("general product form", dict(fold=1, pers=1), {"MIOSQP_PERS_GENERAL": "1"}),
                      ("factor form, sweeps", dict(fold=0, pers=1), {}), ("factor form, S^-1", dict(fold=0, pers=2), {})):
    os.environ.update(env)
    g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, coop=0, resident=0, **kw, **problems.QP_SETTINGS)
    for k in env:
        del os.environ[k]
    r = qp.OSQP(); r.setup(pr["P"], pr["q"], A, l, u, coop=0, resident=0, pers=0, fold=kw["fold"], **problems.QP_SETTINGS)
    for e in (g, r):
        e.set_integer_rows(pr["i_idx"], m)
    assert g.factor_stats()["pers"]
    rng = np.random.RandomState(3)
    x0, y0 = np.zeros(n), np.zeros(M)
    lo, hi = l.copy(), u.copy()
    t0, its = time.time(), 0
    for k in range(nodes):
        a, b = g.solve_node(lo, hi, x0, y0), r.solve_node(lo, hi, x0, y0)
        its += a.iter
        ok = (a.status_val, a.iter) == (b.status_val, b.iter)
        if ok and a.status_val in (1, -2):
            ok = rel(a.x, b.x) <= 1e-9 and rel(a.y, b.y) <= 1e-9
        if not ok:
            bad += 1
            print("MISMATCH %s node %d: status %d/%d iter %d/%d" % (name, k, a.status_val, b.status_val, a.iter, b.iter))
        # walk down a random branch; restart from the root when the relaxation becomes infeasible
        if b.status_val == 1 and k % 25 != 24:
            j = int(rng.randint(250))
            lo, hi = lo.copy(), hi.copy()
            lo[m + j] = hi[m + j] = float(rng.randint(2))
            x0, y0 = b.x, b.y
        else:
            lo, hi, x0, y0 = l.copy(), u.copy(), np.zeros(n), np.zeros(M)
    print("%-22s %d nodes, %d iterations (%d exchanges), %.1f s, mismatches so far %d" %
          (name, nodes, its, its * (2 if kw["fold"] else 4), time.time() - t0, bad), flush=True)
    g.close(); r.close()
# (b) whole searches
for kw in (dict(fold=1, pers=1), dict(fold=0, pers=2)):
    res = []
    for pers in (kw["pers"], 0):
        mdl = bnb.MIOSQP()
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS),
                  dict(problems.QP_SETTINGS, coop=0, resident=0, fold=kw["fold"], pers=pers))
        t0 = time.time()
        r = mdl.solve()
        res.append((r.status, mdl.work.iter_num, mdl.work.osqp_iter, r.upper_glob, time.time() - t0))
        mdl.work.solver.close()
    same = res[0][:3] == res[1][:3] and abs(res[0][3] - res[1][3]) <= 1e-9 * max(1.0, abs(res[1][3]))
    bad += 0 if same else 1
    print("search fold=%d: persistent %s in %.2f s | launches %s in %.2f s | %s" % (kw["fold"], res[0][:4], res[0][4], res[1][:4], res[1][4],
                                                                                 "same" if same else "DIFFERENT"), flush=True)
print("total mismatches", bad)
