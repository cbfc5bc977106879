import sys, os, itertools, subprocess, json
sys.path.insert(0,'/root/repo')
if len(sys.argv) > 1:
    import numpy as np
    from miosqp_amd import qp, problems
    pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
    A,l,u = problems.extended(pr)
    g = qp.OSQP(); g.setup(pr['P'],pr['q'],A,l,u, **problems.QP_SETTINGS)
    r = [g.time_kernel(k, 500)[0] for k in (0,1,4)]
    print(os.environ.get('MIOSQP_FOLD_TPR'), ' '.join('%.3f'%v for v in r))
else:
    for cfg in ["256,64,64","128,64,64","64,64,64","256,32,64","256,64,128","256,64,256","256,64,32","256,32,128","128,32,128","256,16,64","256,128,128"]:
        env = dict(os.environ, MIOSQP_FOLD_TPR=cfg)
        subprocess.call([sys.executable, __file__, 'x'], env=env)
