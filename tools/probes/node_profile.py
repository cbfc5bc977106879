"""cProfile of the node-at-a-time loop (host side) at config 2."""
import sys, cProfile, pstats, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import bnb, dist, problems
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
st = dict(problems.BNB_SETTINGS); st['max_iter_bb'] = 10 ** 9
m = bnb.MIOSQP(); m.setup(pr['P'], pr['q'], pr['A'], pr['l'], pr['u'], pr['i_idx'], pr['i_l'], pr['i_u'], st, dict(problems.QP_SETTINGS))
s = dist.ShardedSearch(m)
for _ in range(20): s.step(1, False)
p = cProfile.Profile(); p.enable()
for _ in range(150): s.step(1, False)
p.disable()
pstats.Stats(p).sort_stats('tottime').print_stats(18)
