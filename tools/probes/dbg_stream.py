import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from miosqp_amd import problems, qp, bnb, stream
n, m, p, seed, cols = 20, 40, 10, 1, 64
pr = problems.random_miqp(n, m, p, seed=seed)
A, l, u = problems.extended(pr)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 6)
model = bnb.MIOSQP()
model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS, max_batch=cols))
ref = qp.OSQP(); ref.setup(pr["P"], pr["q"], A, l, u, **problems.QP_SETTINGS); ref.set_integer_rows(pr["i_idx"], m); ref.set_root(l, u, 1e-3, 1e-3)
def obs(search, g):
    if int(g["status_val"]) == -100: return
    s = int(g["slot"]); par = int(search.parent[s])
    nd = search.eng.pool_read_node(s, p)
    if par >= 0:
        ws = search.eng.pool_read_node(par, p, want=("x", "y")); x0, y0 = ws.x, ws.y
    else:
        x0, y0 = np.zeros(n), np.zeros(len(l))
    l2, u2 = l.copy(), u.copy(); l2[m:], u2[m:] = nd.l, nd.u
    r = ref.solve_node(l2, u2, x0, y0)
    bad = r.status_val in (1, -2) and abs(g["lower"] - r.lower) > 1e-9 * max(1, abs(r.lower))
    print("slot", s, "par", par, "st", g["status_val"], r.status_val, "it", g["iter"], r.iter, "lower", g["lower"], r.lower, "BAD" if bad else "")
    if bad:
        dx = np.abs(nd.x - r.x)
        print("  x diff idx", np.where(dx > 1e-9)[0], dx[dx > 1e-9], "int idx", sorted(pr["i_idx"]))
        ii = pr["i_idx"]
        print("  pool x[ii]", nd.x[ii]); print("  node x[ii]", r.x[ii]); print("  lo", nd.l); print("  hi", nd.u)
srch = stream.StreamSearch(model, columns=cols, observer=obs)
srch.run()
print("---- second instance")
rng = np.random.RandomState(seed)
q2 = rng.randn(n)
model.update_vectors(q=q2)
ref2 = qp.OSQP(); ref2.setup(pr["P"], q2, A, l, u, **problems.QP_SETTINGS); ref2.set_integer_rows(pr["i_idx"], m); ref2.set_root(l, u, 1e-3, 1e-3)
ref = ref2
srch.begin_instance()
srch.run()
