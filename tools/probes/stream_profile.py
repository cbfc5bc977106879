"""cProfile of the host side of the streaming search on config 2 (one pool): where the interpreter's time per chunk goes."""
import cProfile, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, bnb, stream
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
model = bnb.MIOSQP()
model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9), dict(problems.QP_SETTINGS, max_batch=256))
srch = stream.StreamSearch(model, columns=256)
rng = np.random.RandomState(12345)
def steps(k):
    for _ in range(k):
        if srch.step() == 0:
            model.update_vectors(q=rng.randn(500), l=-2 + rng.rand(1000), u=2 + rng.rand(1000))
            srch.begin_instance()
steps(600)
n0, t0 = srch.nodes, time.perf_counter()
p = cProfile.Profile(); p.enable()
steps(400)
p.disable()
dt = time.perf_counter() - t0
print("400 chunks: %.3f s, %d nodes (%.1f us of wall per node)" % (dt, srch.nodes - n0, 1e6 * dt / max(1, srch.nodes - n0)))
pstats.Stats(p).sort_stats("tottime").print_stats(14)
