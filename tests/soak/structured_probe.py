"""Parity of every single-node engine form against the oracle on the structured instances of tests/structured_problems.py
(no assertions: prints what the set-up guard of the explicit inverse measures and how far the iterates are from the
oracle's).  python tools/probes/structured_probe.py [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import structured_problems as sp
from miosqp_amd import problems, qp
from oracle import oracle


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


out = {}
forms = [("coop_unguarded", dict(coop=1, resident=0), {"MIOSQP_GUARD_TOL": "-1"}),
         ("coop", dict(coop=1, resident=0), {}),
         ("launches", dict(coop=0, resident=0, pers=0), {}),
         ("pers1", dict(coop=0, resident=0, pers=1), {})]
names = sys.argv[2:] or list(sp.CASES)
for name in names:
    pr = sp.make(name)
    A, l, u = problems.extended(pr)
    n, M = A.shape[1], A.shape[0]
    o = oracle.OSQP()
    o.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS)
    rng = np.random.RandomState(5)
    x0, y0 = 0.1 * rng.randn(n), 0.1 * rng.randn(M)
    ref = {}
    for k in (1, 2, 27, 75, 200):
        o.warm_start(x=x0, y=y0)
        o.iterate(k)
        ref[k] = [v.copy() for v in o.iterates()]
    o.warm_start(x=np.zeros(n), y=np.zeros(M))
    ro = o.solve()
    rec = dict(n=n, M=M, oracle=(ro.info.status_val, ro.info.iter))
    for fname, kw, env in forms:
        for k2, v in env.items():
            os.environ[k2] = v
        try:
            g = qp.OSQP()
            g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, **kw))
            g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
            fs = g.factor_stats()
            r = dict(coop=fs["coop"], pers=fs["pers"], fold=fs["fold"], guard=g.inverse_guard())
            errs = {}
            for k in (1, 2, 27, 75, 200):
                g.warm_start(x=x0, y=y0)
                xg, zg, yg = g.debug_iterate(k)
                xo, zo, yo = ref[k]
                errs[k] = max(rel(xg, xo), rel(zg, zo), rel(yg, yo))
            r["iter_err"] = errs
            g.warm_start(x=np.zeros(n), y=np.zeros(M))
            rg = g.solve()
            r["solve"] = (rg.info.status_val, rg.info.iter)
            if rg.info.status_val == ro.info.status_val and rg.info.status_val in (1, -2):
                r["sol_err"] = (rel(rg.x, ro.x), rel(rg.y, ro.y))
        except Exception as ex:  # noqa: BLE001
            r = dict(error=repr(ex))
        for k2 in env:
            del os.environ[k2]
        rec[fname] = r
        print(name, fname, json.dumps(r), flush=True)
    out[name] = rec
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1, default=str)
