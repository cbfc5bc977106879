"""us/iter of the cooperative solver for several problem sizes and nap lengths (one subprocess per point)."""
import sys, os, time, subprocess
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1 and sys.argv[1] == 'x':
    import numpy as np
    from miosqp_amd import qp, problems
    n, m, p = (int(v) for v in sys.argv[2:5])
    pr = problems.random_miqp(n, m, p, seed=0); A, l, u = problems.extended(pr)
    M = A.shape[0]
    g = qp.OSQP(); g.setup(pr['P'], pr['q'], A, l, u, coop=1, resident=0, **problems.QP_SETTINGS)
    g.warm_start(x=np.zeros(n), y=np.zeros(M)); g.debug_iterate(10)
    out = []
    for k in (200, 2200):
        t = time.perf_counter(); g.debug_iterate(k); out.append(time.perf_counter() - t)
    print('%.3f' % ((out[1] - out[0]) / 2000 * 1e6), end=' ', flush=True)
else:
    naps = sys.argv[1].split(',')
    for (n, m, p) in [(50, 100, 10), (150, 300, 60), (300, 600, 120), (400, 800, 200), (500, 1000, 250), (580, 1160, 290)]:
        print('N %5d :' % (n + m + p), end=' ', flush=True)
        for nap in naps:
            subprocess.call([sys.executable, __file__, 'x', str(n), str(m), str(p)], env=dict(os.environ, MIOSQP_COOP_NAP=nap + ',1'))
        print(flush=True)
