import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import problems, qp
from oracle import oracle
pr = problems.random_miqp(40, 60, 20, seed=7)
A, l, u = problems.extended(pr)
n, M = 40, A.shape[0]
for fold in (0, 1):
    g = qp.OSQP(); g.setup(pr["P"], pr["q"], A, l, u, fold=fold, resident=0, coop=0, pers=1, **problems.QP_SETTINGS)
    g.set_integer_rows(pr["i_idx"], 60)
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    r = g.solve()
    print("fold", fold, "solve: status", r.info.status_val, "iter", r.info.iter, "|x|", np.max(np.abs(r.x)), "pri", r.info.pri_res, "dua", r.info.dua_res)
    r = g.solve_node(l, u, np.zeros(n), np.zeros(M))
    print("   solve_node: status", r.status_val, "iter", r.iter, "|x|", np.max(np.abs(r.x)), "lower", r.lower)
    g.set_root(l, u, 1e-3, 1e-3)
    r = g.solve_node(l, u, np.zeros(n), np.zeros(M))
    print("   solve_node after set_root: status", r.status_val, "iter", r.iter, "|x|", np.max(np.abs(r.x)), "lower", r.lower, "digest", r.digest.int_inf if r.digest else None)
