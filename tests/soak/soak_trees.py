"""Soak of the whole-tree kernels: random small MIQPs (n 4..60, m 0..90, p 1..n, both exploration rules), MIOSQP.solve on
the HIP engine against the same on the CPU oracle: status, optimum, integer part of x; a few update_vectors re-solves
each.  usage: soak_trees.py [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems  # noqa: E402
from oracle import oracle  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.RandomState(2024)
bad = 0
t0 = time.time()
forms = {}
for k in range(count):
    n = int(rng.randint(4, 61))
    m = int(rng.randint(1, 91))
    p = int(rng.randint(1, min(n, 12) + 1))
    rule = int(rng.randint(0, 2))
    pr = problems.random_miqp(n, m, p, seed=1000 + k)
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, **({"device_tree": False} if os.environ.get("SOAK_HOSTED") else {}))
    a, b = bnb.MIOSQP(), bnb.MIOSQP(backend=oracle)
    for mdl in (a, b):
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                  dict(problems.QP_SETTINGS))
    for step in range(3):
        ra, rb = a.solve(), b.solve()
        ok = ra.status == rb.status
        if ok and np.isfinite(rb.upper_glob):
            ok = abs(ra.upper_glob - rb.upper_glob) <= 1e-6 * max(1.0, abs(rb.upper_glob)) and \
                np.array_equal(np.round(ra.x[pr["i_idx"]]), np.round(rb.x[pr["i_idx"]]))
        if ok:
            ok = a.work.iter_num == b.work.iter_num
        if not ok:
            bad += 1
            print("MISMATCH case %d (n %d m %d p %d rule %d) step %d: %s/%s upper %r/%r nodes %d/%d" %
                  (k, n, m, p, rule, step, ra.status, rb.status, ra.upper_glob, rb.upper_glob, a.work.iter_num, b.work.iter_num))
        q2 = rng.randn(n)
        a.update_vectors(q=q2)
        b.update_vectors(q=q2)
    fs = a.work.solver.factor_stats()
    key = ("tree" if not getattr(a.work, "_no_tree", False) else "hosted", "resident" if fs["resident"] else "coop" if fs["coop"] else "other")
    forms[key] = forms.get(key, 0) + 1
    a.work.solver.close()
print("%d cases x 3 solves, %d mismatches, %.1f s, forms %s" % (count, bad, time.time() - t0, forms))
