"""Where a node-at-a-time step goes: device loop, device total (events), C call, Python around it."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import qp, problems, bnb
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
st = dict(problems.BNB_SETTINGS, device_search=False, device_tree=False); st['max_iter_bb'] = 150  # (the Python loop: solve_node per node)
m = bnb.MIOSQP(); m.setup(pr['P'], pr['q'], pr['A'], pr['l'], pr['u'], pr['i_idx'], pr['i_l'], pr['i_u'], st, dict(problems.QP_SETTINGS))
eng = m.work.solver
calls = []
orig = eng.solve_node
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); calls.append((time.perf_counter() - t, r.info.device_time, r.iter)); return r
eng.solve_node = timed
eng.loop_stats(reset=True)
t = time.perf_counter(); res = m.solve(); wall = time.perf_counter() - t
ms, it = eng.loop_stats()
c = np.array(calls[5:])
n = len(c)
print('nodes', len(calls), 'wall/node %.1f us' % (wall / len(calls) * 1e6))
print('per node (after warm-up): C call %.1f us | device events %.1f us | device loop %.1f us | iters %.0f' % (
    c[:, 0].mean() * 1e6, c[:, 1].mean() * 1e6, ms * 1e3 / len(calls), c[:, 2].mean()))
print('python outside the call: %.1f us/node' % ((wall - sum(x[0] for x in calls)) / len(calls) * 1e6))
