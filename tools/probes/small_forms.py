"""resident (one workgroup, LDS) vs cooperative (registers, exchange) on small problems: whole-solve us/iter"""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from miosqp_amd import qp, problems
def run(P, q, A, l, u, st, **kw):
    n, M = A.shape[1], A.shape[0]
    g = qp.OSQP(); g.setup(P, q, A, l, u, **kw, **st)
    fs = g.factor_stats()
    g.loop_stats(reset=True)
    t = time.perf_counter()
    for rep in range(20):
        g.warm_start(x=np.zeros(n), y=np.zeros(M)); r = g.solve()
    wall = (time.perf_counter() - t) / 20
    ms, it = g.loop_stats()
    return 'res %d coop %d: iter %4d  loop %.3f us/iter  wall %.1f us/solve' % (fs['resident'], fs['coop'], r.info.iter, 1e3 * ms / max(it, 1), wall * 1e6)
for (n, m, p) in [(10, 20, 5), (25, 50, 10), (50, 100, 10), (70, 140, 30)]:
    pr = problems.random_miqp(n, m, p, seed=0); A, l, u = problems.extended(pr)
    print('N', n + A.shape[0])
    for kw in (dict(resident=1, coop=0), dict(resident=0, coop=1), dict(resident=0, coop=0)):
        try: print('  ', run(pr['P'], pr['q'], A, l, u, problems.QP_SETTINGS, **kw))
        except Exception as e: print('  ', kw, 'failed', e)
from golden_cases import load_power_converter
from miosqp_amd import bnb
pc = load_power_converter()
A2, l2, u2 = bnb.add_bounds(pc['i_idx'], pc['i_l'], pc['i_u'], pc['A'], pc['l'], pc['u'][0])
print('power converter N', A2.shape[0] + A2.shape[1])
for kw in (dict(resident=1, coop=0), dict(resident=0, coop=1)):
    try: print('  ', run(pc['P'], pc['q'][0], A2, l2, u2, pc['qp_settings'], **kw))
    except Exception as e: print('  ', kw, 'failed', e)
