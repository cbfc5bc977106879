import sys, os, subprocess
sys.path.insert(0,'/root/repo')
if len(sys.argv) > 1:
    import numpy as np
    from miosqp_amd import qp, problems
    pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
    A,l,u = problems.extended(pr)
    g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u, max_batch=256, **problems.QP_SETTINGS)
    g.set_integer_rows(pr['i_idx'], 1000)
    B=256
    L=np.stack([l]*B); U=np.stack([u]*B); X=np.zeros((B,500)); Y=np.zeros((B,A.shape[0]))
    g.settings.max_iter
    import ctypes
    r = g.solve_batch(L[:B],U[:B],X,Y)
    print(os.environ.get('MIOSQP_BD_CFG'), 'iters', r.iter[0], 'fwd %.1f us  bwd %.1f us  iter %.1f us'%tuple(g.time_kernel(k, 30)[0] for k in (10,11,14)))
else:
    for cfg in ["44","0"]:
        subprocess.call([sys.executable, __file__, 'x'], env=dict(os.environ, MIOSQP_BD_CFG=cfg))
