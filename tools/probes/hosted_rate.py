"""Wall-clock rate of the hosted node-at-a-time search at config 2 (no bench machinery): nodes/s, iterations/s and
microseconds per node outside the iterations (tools/probes; run on the GPU box).  usage: hosted_rate.py [nodes] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems, search  # noqa: E402

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
m = bnb.MIOSQP()
m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st, dict(problems.QP_SETTINGS))
hs = search.HostedSearch(m)
rng = np.random.RandomState(1)
n_, m_ = problems.CONFIGS["cfg2"]["n"], problems.CONFIGS["cfg2"]["m"]


def go(count):
    left = count
    while left > 0:
        before = hs.nodes
        if hs.step(nodes=left) == 0:  # the tree closed: the next MIQP on the same factor
            m.update_vectors(q=rng.randn(n_), l=-2 + rng.rand(m_), u=2 + rng.rand(m_))
            hs.begin_instance()
        left -= max(1, hs.nodes - before)


go(20)
for r in range(reps):
    n0, i0 = hs.nodes, hs.iters
    t0 = time.perf_counter()
    go(nodes)
    dt = time.perf_counter() - t0
    dn, di = hs.nodes - n0, hs.iters - i0
    print("rep %d: %d nodes %d iters  %.1f nodes/s  %.0f it/s  %.1f us/node  it/node %.1f" %
          (r, dn, di, dn / dt, di / dt, 1e6 * dt / dn, di / dn))
