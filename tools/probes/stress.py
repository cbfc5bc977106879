"""Soak: many cooperative launches back to back (small trees solved repeatedly, a long config-2 node chain),
checking determinism and that no launch ever expires."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import bnb, problems, qp
t0 = time.time()
pr = problems.random_miqp(**problems.CONFIGS['cfg1'], seed=0)
ref = None
for rep in range(150):
    m = bnb.MIOSQP(); m.setup(pr['P'], pr['q'], pr['A'], pr['l'], pr['u'], pr['i_idx'], pr['i_l'], pr['i_u'],
                              dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS)) if rep % 50 == 0 else None
    if rep % 50 == 0: model = m
    r = model.solve()
    sig = (r.upper_glob, r.osqp_iter_avg, tuple(np.round(r.x, 12)))
    ref = ref or sig
    assert sig == ref, rep
print('cfg1: 150 solves identical, %.1f s' % (time.time() - t0))
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0); A, l, u = problems.extended(pr)
g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, **problems.QP_SETTINGS); g.set_integer_rows(pr['i_idx'], pr['A'].shape[0])
rng = np.random.RandomState(0)
x, y = np.zeros(500), np.zeros(A.shape[0]); its = 0; t1 = time.time()
for k in range(1500):
    lo, hi = l.copy(), u.copy()
    j = pr['A'].shape[0] + rng.randint(250); v = float(rng.randint(2)); lo[j] = hi[j] = v
    r = g.solve_node(lo, hi, x, y); its += r.iter
    if r.status_val in (1, -2): x, y = r.x, r.y
print('cfg2: 1500 node relaxations, %d iterations, %.1f s, stats %s' % (its, time.time() - t1, g.factor_stats()['coop']))
