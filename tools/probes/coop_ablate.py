"""us/iter of the cooperative solver under env-selected variants (one subprocess per variant)."""
import sys, os, time, subprocess
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == 'x':
    import numpy as np
    from miosqp_amd import qp, problems
    pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0); A, l, u = problems.extended(pr)
    n, M = A.shape[1], A.shape[0]
    g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, coop=1, **problems.QP_SETTINGS)
    g.warm_start(x=np.zeros(n), y=np.zeros(M)); g.debug_iterate(10)
    out = []
    for k in (200, 2200):
        t = time.perf_counter(); g.debug_iterate(k); out.append(time.perf_counter() - t)
    print(sys.argv[2], 'us/iter %.3f' % ((out[1] - out[0]) / 2000 * 1e6), flush=True)
else:
    for name, env in [(a, dict(kv.split('=') for kv in a.split())) for a in sys.argv[1:]]:
        subprocess.call([sys.executable, __file__, 'x', name], env=dict(os.environ, **env))
