"""Soak of the streaming search: the compiled host driver on one pool, then two pools sharing trees, for about two
minutes of MIQP after MIQP at config 2 (256 columns); every tree must close with every slot returned (the driver
refuses a digest for a slot that is not in flight), and the hosted node-at-a-time search on the same instances must
agree on the optimum of a sample."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from miosqp_amd import bnb, problems, stream  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
cfg = problems.CONFIGS["cfg2"]
pr = problems.random_miqp(**cfg, seed=0)
st = dict(problems.BNB_SETTINGS, max_iter_bb=3000)


def make():
    m = bnb.MIOSQP()
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], dict(st),
            dict(problems.QP_SETTINGS, max_batch=256))
    return m


model = make()
ns = stream.NativeStreamSearch(model, columns=256)
rng = np.random.RandomState(7)
t0, trees, nodes = time.time(), 0, 0
while time.time() - t0 < budget:
    r = ns.run()
    assert len(ns.free) == ns.capacity or r.status != bnb.MI_SOLVED, (trees, len(ns.free))
    trees += 1
    model.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(cfg["m"]), u=2 + rng.rand(cfg["m"]))
    ns.begin_instance()
print("one pool: %d trees, %d nodes, %.1f s, persistent launches called off %d" %
      (trees, ns.nodes, time.time() - t0, model.work.solver.batch_pers_fallbacks()))
model.work.solver.close()

mp = stream.MultiPoolSearch(make, pools=2, columns=256, exchange_every=4, driver="native")
rng = np.random.RandomState(8)
t0, trees = time.time(), 0
while time.time() - t0 < budget:
    r = mp.run()
    assert all(len(sh.ss.free) == sh.ss.capacity for sh in mp.sh) or r.status != bnb.MI_SOLVED
    trees += 1
    mp.update_vectors(q=rng.randn(cfg["n"]), l=-2 + rng.rand(cfg["m"]), u=2 + rng.rand(cfg["m"]))
fb = [m.work.solver.batch_pers_fallbacks() for m in mp.models]
print("two pools: %d trees, %d nodes, %.1f s, persistent launches called off %s, whole-chip launches that took turns %d" %
      (trees, sum(sh.ss.nodes for sh in mp.sh), time.time() - t0, fb, mp.models[0].work.solver.chip_turn_waits()))
assert sum(fb) == 0, fb
