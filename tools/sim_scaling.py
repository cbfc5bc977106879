#!/usr/bin/env python
"""Leaf-sharding simulator: W ranks as threads of one process over the CPU oracle, with a model clock
(per node: iterations * T_ITER + T_NODE; per exchange: T_SYNC and a max over ranks), to choose the sharding
parameters (nodes per rank between incumbent exchanges) before the multi-GPU bench runs.  Not a benchmark.

    python tools/sim_scaling.py [cfg] [seed] [instances]
"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import digest_backend
from miosqp_amd import bnb, dist, problems

import os
os.environ.setdefault("MIOSQP_EXCHANGE_LAG", "0")  # the model clock has its own notion of lag (SIM_LAG)
T_ITER, T_NODE, T_SYNC = 2.83e-6, 170e-6, 60e-6
LAG = int(os.environ.get("SIM_LAG", "0"))  # model of a non-blocking exchange: wait only for the one LAG steps back


class SimWorld(object):
    def __init__(self, world):
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.clock = [0.0] * world
        self.busy = [0.0] * world
        self.hist = []  # per exchange: the latest arrival over ranks


class SimComm(object):
    def __init__(self, sw, rank):
        self.sw, self.rank, self.world = sw, rank, sw.world
        self.step_no = 0

    def _all(self, obj):
        sw = self.sw
        sw.slots[self.rank] = obj
        sw.bar.wait()
        out = list(sw.slots)
        sw.bar.wait()
        return out

    def exchange(self, value, x, nleaves, have=None):
        sw = self.sw
        tab = self._all((value, nleaves, x, sw.clock[self.rank]))
        if self.rank == 0:
            sw.hist.append(max(t[3] for t in tab))
        sw.bar.wait()
        k = self.step_no - LAG
        self.step_no += 1
        if LAG == 0:
            sw.clock[self.rank] = sw.hist[-1] + T_SYNC
        else:
            sw.clock[self.rank] = max(sw.clock[self.rank], sw.hist[k] if k >= 0 else 0.0) + 0.25 * T_SYNC
        self._counts = [int(t[1]) for t in tab]
        vals = np.array([t[0] for t in tab])
        owner = int(np.argmin(vals))
        best = float(vals[owner])
        total = sum(self._counts)
        prev = float(np.max(vals)) if have is None else have
        if not np.isfinite(best) or not best < prev:
            return best, owner, None, total
        return best, owner, np.array(tab[owner][2]), total

    def post(self, value, x, nleaves):
        return (value, None if x is None else np.array(x), nleaves)

    def complete(self, h, have=None):
        return self.exchange(h[0], h[1], h[2], have)

    def leaf_counts(self):
        return list(self._counts)

    def move(self, arr, size, src):
        tab = self._all(None if arr is None else np.array(arr))
        return np.array(tab[src])

    def sum(self, arr):
        tab = self._all(np.asarray(arr, dtype=np.float64))
        return np.sum(tab, axis=0)

    def barrier(self):
        self.sw.bar.wait()


def rank_main(sw, rank, prob, wave, instances, seed, out):
    st = dict(problems.BNB_SETTINGS); st["max_iter_bb"] = 10 ** 9
    m = bnb.MIOSQP(backend=digest_backend)
    m.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], st,
            dict(problems.QP_SETTINGS))
    comm = SimComm(sw, rank)
    s = dist.ShardedSearch(m, comm)
    orig = s._visit

    def visit(rule):
        leaf = orig(rule)
        dt = leaf.num_iter * T_ITER + T_NODE
        sw.clock[rank] += dt
        sw.busy[rank] += dt
        return leaf
    s._visit = visit
    s.clock = lambda: sw.clock[rank]
    budget = float(os.environ.get("SIM_BUDGET", "0")) or None
    rng = np.random.RandomState(seed + 12345)
    cfgm = prob["A"].shape[0]
    for inst in range(instances):
        while s.step(wave, False, budget=budget) != 0:
            pass
        q = rng.randn(prob["A"].shape[1]); u = 2 + rng.rand(cfgm); l = -2 + rng.rand(cfgm)
        m.update_vectors(q=q, l=l, u=u)
        s.begin_instance()
    out[rank] = (s.nodes, s.iters)


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    instances = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prob = problems.random_miqp(**problems.CONFIGS[cfg], seed=seed)
    base = None
    print("cfg %s seed %d instances %d; model: %.2f us/iter, %.0f us/node, %.0f us/exchange" % (
        cfg, seed, instances, T_ITER * 1e6, T_NODE * 1e6, T_SYNC * 1e6))
    worlds = [int(v) for v in os.environ.get("SIM_WORLDS", "1,2,4,8").split(",")]
    waves = [int(v) for v in os.environ.get("SIM_WAVES", "1,4,16").split(",")]
    for world in worlds:
        for wave in ((1,) if world == 1 else waves):
            sw = SimWorld(world)
            out = [None] * world
            th = [threading.Thread(target=rank_main, args=(sw, r, prob, wave, instances, seed, out)) for r in range(world)]
            t0 = time.time()
            [t.start() for t in th]; [t.join() for t in th]
            nodes = sum(o[0] for o in out); iters = sum(o[1] for o in out)
            T = max(sw.clock)
            rate = iters / T
            if base is None:
                base = rate
            print("world %d wave %2d: nodes %5d iters %8d  model time %.3f s  %.0f iter/s  efficiency %.2f  busy %.2f  (%.0f s cpu)" % (
                world, wave, nodes, iters, T, rate, rate / (world * base), sum(sw.busy) / (world * T), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
