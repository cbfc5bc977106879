"""Soak of the hosted node-at-a-time search in the cooperative range (193 <= n + M <= 2048: one k_coop launch per node,
prologue and epilogue inside it) against the SAME search on the CPU oracle (bnb.MIOSQP(backend=oracle): the Python loop
of solver.py:65-172 on the CPU restatement): random shapes, both exploration rules, rho = 0.1 and rho = "auto", the first
`max_iter_bb` nodes of each tree: status, nodes, ADMM iterations, incumbent value, integer part of the incumbent.
usage: soak_hosted_coop.py [count] [max_iter_bb]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems  # noqa: E402
from oracle import oracle  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.RandomState(4242)
bad, t0, forms, nodes, iters = 0, time.time(), {}, 0, 0
for k in range(count):
    n = int(rng.randint(60, 260))
    m = int(rng.randint(max(1, 193 - 2 * n), 520))
    p = int(rng.randint(2, min(n, 24) + 1))
    rule = int(rng.randint(0, 2))
    rho = "auto" if k % 3 == 2 else 0.1
    pr = problems.random_miqp(n, m, p, seed=9000 + k, density=float(rng.choice([0.1, 0.4, 0.7])))
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, max_iter_bb=cap, device_tree=False)
    qs = dict(problems.QP_SETTINGS, rho=rho)
    a, b = bnb.MIOSQP(), bnb.MIOSQP(backend=oracle)
    for mdl in (a, b):
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(qs))
    fs = a.work.solver.factor_stats()
    key = "coop" if fs["coop"] else "resident" if fs["resident"] else "other"
    forms[key] = forms.get(key, 0) + 1
    ra, rb = a.solve(), b.solve()
    hosted = getattr(a.work, "_hosted", None) is not None
    same = (ra.status == rb.status and a.work.iter_num == b.work.iter_num and a.work.osqp_iter == b.work.osqp_iter)
    if same and np.isfinite(rb.upper_glob):
        same = abs(ra.upper_glob - rb.upper_glob) <= 1e-8 * max(1.0, abs(rb.upper_glob)) and \
            np.array_equal(np.round(ra.x[pr["i_idx"]]), np.round(rb.x[pr["i_idx"]]))
    nodes += a.work.iter_num
    iters += a.work.osqp_iter
    if not same or not hosted:
        bad += 1
        print("MISMATCH case %d n=%d m=%d p=%d rule=%d rho=%s hosted=%s: hip (%s, %d, %d, %r) oracle (%s, %d, %d, %r)" %
              (k, n, m, p, rule, rho, hosted, ra.status, a.work.iter_num, a.work.osqp_iter, ra.upper_glob,
               rb.status, b.work.iter_num, b.work.osqp_iter, rb.upper_glob))
print("%d searches (first %d nodes each), %d nodes, %d ADMM iterations, %d mismatches, %.1f s, forms %s" %
      (count, cap, nodes, iters, bad, time.time() - t0, forms))
