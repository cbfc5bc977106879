"""k_coop's row layouts side by side on one problem: the full layout (identity rows in W), the reduced layout with 8 and
with 16 rows per workgroup -- microseconds per iteration back to back (long debug_iterate runs) and the hosted
node-at-a-time search's rate.  usage: coop_layout.py [n m p] [nodes]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems, qp, search  # noqa: E402

if len(sys.argv) >= 4:
    n, m, p = (int(v) for v in sys.argv[1:4])
    nodes = int(sys.argv[4]) if len(sys.argv) > 4 else 150
else:
    c = problems.CONFIGS["cfg2"]
    n, m, p = c["n"], c["m"], c["p"]
    nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pr = problems.random_miqp(n, m, p, seed=0)
A, l, u = problems.extended(pr)
M = A.shape[0]
for name, env in (("full layout, 8 rows", {"MIOSQP_COOP_IDROWS": "0", "MIOSQP_COOP_RW": "8"}),
                  ("reduced, launches per node", {"MIOSQP_COOP_NODE": "0"}),
                  ("reduced, prologue in", {"MIOSQP_COOP_EPI": "0"}),
                  ("reduced, one launch", {})):
    for k, v in env.items():
        os.environ[k] = v
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], m)
    fs = g.factor_stats()
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    out = []
    for k in (200, 2200, 200, 2200):
        t = time.perf_counter()
        g.debug_iterate(k)
        out.append(time.perf_counter() - t)
    us = min(out[1] - out[0], out[3] - out[2]) / 2000 * 1e6
    g.close()
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
    mm = bnb.MIOSQP()
    mm.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st, dict(problems.QP_SETTINGS))
    hs = search.HostedSearch(mm)
    rng = np.random.RandomState(1)

    def go(count):
        left = count
        while left > 0:
            before = hs.nodes
            if hs.step(nodes=left) == 0:
                mm.update_vectors(q=rng.randn(n), l=-2 + rng.rand(m), u=2 + rng.rand(m))
                hs.begin_instance()
            left -= max(1, hs.nodes - before)

    go(20)
    best = None
    for r in range(2):
        n0, i0 = hs.nodes, hs.iters
        t0 = time.perf_counter()
        go(nodes)
        dt = time.perf_counter() - t0
        dn, di = hs.nodes - n0, hs.iters - i0
        rec = (di / dt, dn / dt, 1e6 * dt / dn, di / dn)
        if best is None or rec[0] > best[0]:
            best = rec
    print("%-28s coop %s nap %2d: %.3f us/iter back to back | hosted: %.0f it/s  %.1f nodes/s  %.1f us/node  %.1f it/node"
          % (name, fs["coop"], fs["coop_nap"], us, best[0], best[1], best[2], best[3]), flush=True)
    mm.work.solver.close()
    for k in env:
        del os.environ[k]
