import os, sys, time, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, bnb, stream, dist
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
# sequential: when does the first incumbent appear?
m0 = bnb.MIOSQP()
m0.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS))
rows = []
m0.solve(observer=lambda w, lf: rows.append((lf.depth, lf.status, lf.num_iter, lf.intinf, w.upper_glob, lf.lower, lf.digest.heur_feasible if lf.digest else None)))
first = next(i for i, r in enumerate(rows) if np.isfinite(r[4]))
print("sequential: %d nodes, first incumbent at node %d depth %d; statuses %s" % (len(rows), first, rows[first][0], collections.Counter(r[1] for r in rows)))
print(" first 12 rows", rows[:12])
model = bnb.MIOSQP()
model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st), dict(problems.QP_SETTINGS, max_batch=256))
stat = collections.Counter(); its = []; hv = []; ii = []; dep = []
def obs(s, g):
    stat[int(g["status_val"])] += 1; its.append(int(g["iter"])); hv.append(float(g["heur_viol"])); ii.append(int(g["int_inf"])); dep.append(int(s.depth[int(g["slot"])]))
srch = stream.StreamSearch(model, columns=256, observer=obs)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 320):
    srch.step()
    if k % 200 == 0: print(' step', k, 'nodes', srch.nodes, 'open', len(srch.open), 'free', len(srch.free), 'maxdepth', max(dep) if dep else 0, 'upper', model.work.upper_glob, 'active', srch.active)
print("stream: nodes", srch.nodes, "status", stat, "iters med", np.median(its), "min intinf", min(x for x in ii if x >= 0), "heur ok", sum(1 for v in hv if v <= 0), "max depth", max(dep), "upper", model.work.upper_glob)
print(" first 12:", list(zip(dep, its, ii, [round(v, 4) for v in hv]))[:12])
