"""Soak of the streaming search over random shapes (odd and even sizes): the compiled host driver and the Python one on
the leaf pool against the oracle's sequential search: status and optimum, every slot returned.  usage: [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import bnb, problems, stream  # noqa: E402
from oracle import oracle  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(99)
bad, t0 = 0, time.time()
for k in range(count):
    n = int(rng.randint(5, 70))
    m = int(rng.randint(3, 130))
    p = int(rng.randint(2, min(n, 14) + 1))
    rule = int(rng.randint(0, 2))
    cols = int(rng.choice([64, 128]))
    pr = problems.random_miqp(n, m, p, seed=9000 + k)
    st = dict(problems.BNB_SETTINGS, tree_explor_rule=rule, max_iter_bb=10 ** 6)
    ref = bnb.MIOSQP(backend=oracle)
    ref.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS))
    r0 = ref.solve()
    for kind in ("native", "python"):
        mdl = bnb.MIOSQP()
        mdl.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
                  dict(problems.QP_SETTINGS, max_batch=cols))
        s = (stream.NativeStreamSearch if kind == "native" else stream.StreamSearch)(mdl, columns=cols, capacity=4096)
        r1 = s.run()
        ok = r1.status == r0.status and len(s.free) == s.capacity
        if ok and np.isfinite(r0.upper_glob):
            ok = abs(r1.upper_glob - r0.upper_glob) <= 1e-3 * max(1.0, abs(r0.upper_glob)) and \
                np.array_equal(np.round(r1.x[pr["i_idx"]]), np.round(r0.x[pr["i_idx"]]))
        if not ok:
            bad += 1
            print("MISMATCH case %d (n %d m %d p %d rule %d cols %d) %s: %s/%s upper %r/%r free %d/%d" %
                  (k, n, m, p, rule, cols, kind, r1.status, r0.status, r1.upper_glob, r0.upper_glob, len(s.free), s.capacity))
        mdl.work.solver.close()
print("%d shapes x 2 drivers, %d mismatches, %.1f s" % (count, bad, time.time() - t0))
