"""cProfile of the batched wave loop (host side) at config 2."""
import sys, time, cProfile, pstats, numpy as np
sys.path.insert(0, '/root/repo')
from miosqp_amd import bnb, dist, problems
pr = problems.random_miqp(**problems.CONFIGS['cfg2'], seed=0)
st = dict(problems.BNB_SETTINGS); st['max_iter_bb'] = 10 ** 9
qs = dict(problems.QP_SETTINGS); qs['max_batch'] = 256
m = bnb.MIOSQP(); m.setup(pr['P'], pr['q'], pr['A'], pr['l'], pr['u'], pr['i_idx'], pr['i_l'], pr['i_u'], st, qs)
s = dist.ShardedSearch(m)
rng = np.random.RandomState(12345)
def next_instance():
    m.update_vectors(q=rng.randn(500), u=2 + rng.rand(1000), l=-2 + rng.rand(1000)); s.begin_instance()
def waves(k):
    for _ in range(k):
        if s.step(256, True) == 0: next_instance()
waves(12)
eng = m.work.solver; eng.batch_stats(reset=True)
n0 = s.nodes; t = time.perf_counter()
pr_ = cProfile.Profile(); pr_.enable(); waves(12); pr_.disable()
dt = time.perf_counter() - t; ms, bi, ni = eng.batch_stats()
print('12 waves: %.1f ms wall, %.1f ms device loop, nodes %d' % (dt * 1e3, ms, s.nodes - n0))
pstats.Stats(pr_).sort_stats('cumulative').print_stats(22)
