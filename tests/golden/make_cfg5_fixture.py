"""Generates tests/golden/cfg5_root.npz: BASELINE config 5 (random_miqp n=5000 m=10000 p=2500, 1 % dense A,
seed 0) solved by the CPU oracle (oracle/qp_oracle.c) at the root and at the root's two children, in the call
order of /root/reference/miosqp/node.py:102-143 (update bounds, warm start, solve, integer clamp, objective).

Build container only (the oracle needs minutes at this size, which is why the GPU test reads this fixture
instead of running it):  python tests/golden/make_cfg5_fixture.py

What is stored: the instance's digest (the GPU test regenerates the instance with the same seeded recipe and
refuses to compare when scipy's sampling has changed), per node status / iterations / x after the integer
clamp / y / lower bound (the objective at the clamped x), and the two children's branching variable and bounds.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from miosqp_amd import problems  # noqa: E402
from oracle import oracle  # noqa: E402


def solve_node(o, pr, l, u, x0, y0):
    o.update(l=l, u=u)
    o.warm_start(x=x0, y=y0)
    r = o.solve()
    x = r.x.copy()
    ii = pr["i_idx"]
    p = len(ii)
    x[ii] = np.minimum(np.maximum(x[ii], l[-p:]), u[-p:])  # node.py:131-136
    lower = 0.5 * x.dot(pr["P"].dot(x)) + pr["q"].dot(x)   # data.py:99-103
    return dict(status=r.info.status_val, iter=r.info.iter, x=x, y=r.y.copy(), lower=lower,
                pri_res=r.info.pri_res, dua_res=r.info.dua_res)


def main():
    cfg = problems.CONFIGS["cfg5"]
    pr = problems.random_miqp(**cfg, seed=0)
    A, l, u = problems.extended(pr)
    n, M, m = cfg["n"], A.shape[0], cfg["m"]
    t0 = time.time()
    o = oracle.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    print("oracle setup %.1f s, nnz(L) %d" % (time.time() - t0, o.factor_nnz()), flush=True)
    t0 = time.time()
    root = solve_node(o, pr, l, u, np.zeros(n), np.zeros(M))
    print("root: status %d iter %d lower %.9g (%.1f s)" % (root["status"], root["iter"], root["lower"],
                                                          time.time() - t0), flush=True)
    xi = root["x"][pr["i_idx"]]
    k = int(np.argmax(np.abs(xi - np.round(xi))))  # the same child pair the property test builds
    lo, up = [l.copy(), l.copy()], [u.copy(), u.copy()]
    up[0][m + k] = np.floor(xi[k])
    lo[1][m + k] = np.ceil(xi[k])
    kids = []
    for c in (0, 1):
        t0 = time.time()
        kids.append(solve_node(o, pr, lo[c], up[c], root["x"], root["y"]))
        print("child %d: status %d iter %d lower %.9g (%.1f s)" % (c, kids[c]["status"], kids[c]["iter"],
                                                                   kids[c]["lower"], time.time() - t0), flush=True)
    out = dict(digest=np.array(problems.instance_digest(pr)), branch_k=np.array(k),
               branch_floor=np.array(np.floor(xi[k])), branch_ceil=np.array(np.ceil(xi[k])))
    for name, r in (("root", root), ("child0", kids[0]), ("child1", kids[1])):
        for key, v in r.items():
            out["%s_%s" % (name, key)] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, "cfg5_root.npz"), **out)
    print("written", os.path.join(HERE, "cfg5_root.npz"))


if __name__ == "__main__":
    main()
