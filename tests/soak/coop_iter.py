"""Cooperative register-resident solver vs the two-kernel product form on config 2: iterate parity
against the oracle, time per iteration (long debug_iterate runs), and whole solves."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from miosqp_amd import qp, problems
from oracle import oracle
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
pr = problems.random_miqp(**problems.CONFIGS[cfg], seed=0); A, l, u = problems.extended(pr)
n, M = A.shape[1], A.shape[0]
o = oracle.OSQP(); o.setup(pr['P'], pr['q'], A, l, u, **problems.QP_SETTINGS)
o.warm_start(x=np.zeros(n), y=np.zeros(M)); o.iterate(60); xo, zo, yo = o.iterates()
ro = o.solve()
for coop in (0, 1):
    g = qp.OSQP(); t = time.perf_counter(); g.setup(pr['P'], pr['q'], A, l, u, coop=coop, **problems.QP_SETTINGS)
    ts = time.perf_counter() - t
    print('coop', coop, g.factor_stats(), 'setup %.3f s' % ts)
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    x, z, y = g.debug_iterate(60)
    print('  iterate(60) max err x %.2e z %.2e y %.2e' % (abs(x - xo).max(), abs(z - zo).max(), abs(y - yo).max()))
    out = []
    for k in (200, 2200):
        t = time.perf_counter(); g.debug_iterate(k); out.append(time.perf_counter() - t)
    print('  us/iter %.3f' % ((out[1] - out[0]) / 2000 * 1e6))
    g.warm_start(x=np.zeros(n), y=np.zeros(M))
    g.loop_stats(reset=True)
    for rep in range(5):
        g.warm_start(x=np.zeros(n), y=np.zeros(M))
        r = g.solve()
    ms, it = g.loop_stats()
    print('  solve: status %d iter %d (oracle %d %d) x err %.2e  loop us/iter %.3f' % (
        r.info.status_val, r.info.iter, ro.info.status_val, ro.info.iter, abs(r.x - ro.x).max(), ms * 1e3 / max(it, 1)))
