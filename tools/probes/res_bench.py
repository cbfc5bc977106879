import sys, time, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from miosqp_amd import qp, problems, bnb
from golden_cases import load_power_converter, run_power_converter
pc = load_power_converter()
for res in (0,1):
    pc2 = dict(pc); pc2["qp_settings"] = dict(pc["qp_settings"], resident=res)
    t=time.time(); out = run_power_converter(pc2, qp); dt=time.time()-t
    nodes=sum(o["nodes"] for o in out); its=sum(o["osqp_iter"] for o in out)
    print("power_converter resident=%d: %d MIQPs, %d nodes, %d iters in %.3fs -> %.0f nodes/s %.0f iter/s"%(res,len(out),nodes,its,dt,nodes/dt,its/dt))
pr = problems.random_miqp(50,100,10,seed=0)
for res in (0,1):
    m = bnb.MIOSQP(); m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS), dict(problems.QP_SETTINGS, resident=res))
    t=time.time(); r=m.solve(); dt=time.time()-t
    ms,it = m.work.solver.loop_stats()
    print("cfg1 resident=%d: status %s nodes %d iters %d in %.4fs; device loop %.3f ms for %d iters = %.3f us/iter"%(res, r.status, m.work.iter_num-1, m.work.osqp_iter, dt, ms, it, 1e3*ms/max(it,1)))
