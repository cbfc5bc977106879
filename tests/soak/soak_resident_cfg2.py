"""Soak of the resident search grid at config-2 size: MIQP after MIQP of the bench's stream (update_vectors with fresh q, l, u on
one factor), every tree closed twice -- the grid resident over search_run calls of random length, and one cooperative launch
per node (MIOSQP_COOP_RUN is read per call) -- and compared for EQUALITY: nodes, ADMM iterations, incumbent value and vector.
usage: soak_resident_cfg2.py [instances] [rho: 0.1 | auto]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems, search  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rho = sys.argv[2] if len(sys.argv) > 2 else "0.1"
cfg = problems.CONFIGS["cfg2"]
pr = problems.random_miqp(seed=0, **cfg)
qs = dict(problems.QP_SETTINGS)
if rho == "auto":
    qs["rho"] = "auto"
models = []
for run in (1, 0):
    m = bnb.MIOSQP()
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9), dict(qs))
    models.append((run, m, search.HostedSearch(m)))
rng = np.random.RandomState(777)
bad, nodes, iters, t0 = 0, 0, 0, time.time()
for k in range(count):
    res = []
    pieces = rng.randint(1, 60)
    for run, m, hs in models:
        os.environ["MIOSQP_COOP_RUN"] = str(run)
        n0, i0 = hs.nodes, hs.iters
        alive = 1
        while alive:
            alive = hs.step(int(pieces) if run else 10 ** 9)
        res.append((hs.nodes - n0, hs.iters - i0, float(m.work.upper_glob), None if m.work.x is None else np.array(m.work.x)))
    a, b = res
    ok = a[:3] == b[:3] and ((a[3] is None and b[3] is None) or np.array_equal(a[3], b[3]))
    if not ok:
        bad += 1
        print("MISMATCH instance %d: resident %r per-node %r" % (k, a[:3], b[:3]))
    nodes += a[0]
    iters += a[1]
    q, u, l = rng.randn(cfg["n"]), 2 + rng.rand(cfg["m"]), -2 + rng.rand(cfg["m"])
    for run, m, hs in models:
        m.update_vectors(q=q, l=l, u=u)
        hs.begin_instance()
fb = [m.work.solver.factor_stats()["coop_fallbacks"] for _, m, _ in models]
print("%d MIQPs at config-2 size (rho %s), %d nodes, %d ADMM iterations, resident grid against a launch per node: %d mismatches, fall-backs %s, %.1f s"
      % (count, rho, nodes, iters, bad, fb, time.time() - t0))
