"""Model of bench.py at N > 1 GPUs WITHOUT the GPUs: W ranks as threads of one process, each running exactly what a rank
of `torchrun ... bench.py --gpus W` runs -- dist.ShardedStream over search.HostedSearch (replicated ramp-up, deal,
node-at-a-time steps, the exchange with its feeding of dry ranks) -- with the CPU restatement doing the relaxations
(tests/digest_backend.py) and a MODEL CLOCK instead of wall time: a node costs its ADMM iterations x T_ITER + T_NODE
(the resident grid's measured figures, r06), a collective makes every rank wait for the slowest and costs T_COLL, a moved
leaf T_LEAF.  What it answers: how much of the machine the leaf-sharded search keeps busy at 2 / 4 / 8 ranks on config 2's
real trees, as a function of the deal's parameters (leaves per rank before the deal, nodes per rank between exchanges,
leaves fed to a dry rank) -- the SCALE curve has never been measured on hardware (no multi-GPU box in any round).
NOT a measurement: the clock is a model.

    python tests/soak/sim_sharded_hosted.py [instances] > profiles/rNN_sim_sharded_hosted.txt
    SIM_WORLDS=1,2,4,8  SIM_SETS="nodes,every,ramp,feed,budget_us;..." (budget_us > 0: steps by time, as bench.py's --step-budget-ms)  (defaults below)
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import digest_backend  # noqa: E402
from miosqp_amd import bnb, dist, poolcomm, problems, search  # noqa: E402

T_ITER, T_NODE = 1.86e-6, 25e-6       # resident grid, config 2 (r06: 1.86 us per iteration in situ, ~25 us per node outside them)
T_COLL, T_LEAF = 60e-6, 40e-6         # one small RCCL collective + its host side; one leaf record broadcast (guesses, stated)


class SimWorld(poolcomm.PoolWorld):
    def __init__(self, world):
        poolcomm.PoolWorld.__init__(self, world)
        self.clock = [0.0] * world
        self.busy = [0.0] * world
        self.colls = 0


class SimComm(poolcomm.PoolComm):
    """PoolComm whose collectives also synchronise the model clocks: everybody leaves at max(arrivals) + cost"""

    def _meet(self, cost):
        tw = self.tw
        tab = poolcomm.PoolComm._all(self, tw.clock[self.rank])
        tw.clock[self.rank] = max(tab) + cost
        if self.rank == 0:
            tw.colls += 1

    def gather(self, vec):
        out = poolcomm.PoolComm.gather(self, vec)
        self._meet(T_COLL)
        return out

    def exchange(self, value, x, nleaves, have=None, extra=(0.0, 0.0)):
        out = poolcomm.PoolComm.exchange(self, value, x, nleaves, have, extra)
        self._meet(T_COLL + (T_COLL if out[2] is not None else 0.0))  # (+ the broadcast of x when somebody improved it)
        return out

    def move(self, arr, size, src):
        out = poolcomm.PoolComm.move(self, arr, size, src)
        self._meet(T_LEAF)
        return out

    def sum(self, arr):
        out = poolcomm.PoolComm.sum(self, arr)
        self._meet(T_COLL)
        return out


def rank_main(sw, rank, prob, pset, instances, seed, out, err):
    try:
        nodes_per_step, every, ramp, feed, budget_us = pset
        st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
        m = bnb.MIOSQP(backend=digest_backend)
        m.setup(prob["P"], prob["q"], prob["A"], prob["l"], prob["u"], prob["i_idx"], prob["i_l"], prob["i_u"], st,
                dict(problems.QP_SETTINGS))
        comm = SimComm(sw, rank)
        hs = search.HostedSearch(m, capacity=4096)
        step0 = hs.step
        idle = [0, 0]

        def step(nodes=None, budget=None):
            n0, i0 = hs.nodes, hs.iters
            if budget_us > 0:
                # bench.py's default at N > 1: node relaxations for a TIME budget (checked between nodes), on the model clock
                alive = 1
                while alive != 0:
                    alive = step0(1)
                    if (hs.iters - i0) * T_ITER + (hs.nodes - n0) * T_NODE >= 1e-6 * budget_us:
                        break
            else:
                alive = step0(nodes)
            dt = (hs.iters - i0) * T_ITER + (hs.nodes - n0) * T_NODE
            sw.clock[rank] += dt
            sw.busy[rank] += dt
            idle[0] += 1
            idle[1] += 1 if hs.nodes == n0 else 0
            return alive
        hs.step = step
        sh = dist.ShardedStream(m, comm, search=hs, exchange_every=every, ramp_leaves=ramp, feed=feed,
                                step_kwargs=dict(nodes=nodes_per_step))
        # (the replicated ramp-up runs on every rank at once: one rank's worth of model time, counted once in the totals)
        seq_visit = sh.seq._visit

        def visit(rule):
            leaf = seq_visit(rule)
            dt = leaf.num_iter * T_ITER + T_NODE
            sw.clock[rank] += dt
            if rank == 0:
                sw.busy[rank] += dt
            return leaf
        sh.seq._visit = visit
        rng = np.random.RandomState(seed + 12345)
        mo = prob["A"].shape[0]
        # (the first instance's ramp-up ran inside the constructor, before the wrapper: charge it now)
        sw.clock[rank] += sh.global_iters * T_ITER + sh.global_nodes * T_NODE
        if rank == 0:
            sw.busy[rank] += sh.global_iters * T_ITER + sh.global_nodes * T_NODE
        tot_n = tot_i = 0
        for inst in range(instances):
            while sh.step() != 0:
                pass
            tot_n += sh.global_nodes
            tot_i += sh.global_iters
            if inst + 1 < instances:
                m.update_vectors(q=rng.randn(prob["A"].shape[1]), l=-2 + rng.rand(mo), u=2 + rng.rand(mo))
                sh.begin_instance()
        out[rank] = dict(nodes=tot_n, iters=tot_i, idle_steps=idle[1], steps=idle[0], moved=sh.moved, upper=float(m.work.upper_glob))
    except Exception as ex:  # noqa: BLE001
        err.append(ex)
        sw.fail(ex)


def main():
    instances = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    seed = 0
    shape = problems.CONFIGS["cfg2"]
    if os.environ.get("SIM_SHAPE"):  # (a quick look at the script itself: SIM_SHAPE=50,100,25)
        n, m, p = [int(v) for v in os.environ["SIM_SHAPE"].split(",")]
        shape = dict(n=n, m=m, p=p, density=0.7)
    prob = problems.random_miqp(seed=seed, **shape)
    worlds = [int(v) for v in os.environ.get("SIM_WORLDS", "1,2,4,8").split(",")]
    sets = [tuple(int(v) for v in s.split(",")) for s in
            os.environ.get("SIM_SETS", "1,1,1,64,0;1,1,1,64,2000;1,1,2,8,4000;1,1,2,8,8000;1,1,4,8,8000").split(";")]
    print("# model of `bench.py --gpus W` (leaf-sharded hosted search) on random_miqp n=%d m=%d p=%d, %d MIQP(s) of the stream, seed %d"
          % (shape["n"], shape["m"], shape["p"], instances, seed))
    print("# model clock: %.2f us per ADMM iteration, %.0f us per node, %.0f us per collective, %.0f us per moved leaf -- NOT a measurement"
          % (T_ITER * 1e6, T_NODE * 1e6, T_COLL * 1e6, T_LEAF * 1e6))
    print("# efficiency = (ADMM iterations of all ranks / model time) / (W x the same at W = 1); idle = steps in which a rank had no leaf")
    base = {}
    for pset in sets:
        for world in worlds:
            if world == 1 and base:  # (one rank takes no part in any exchange: the same run whatever the parameters)
                continue
            sw = SimWorld(world)
            out, err = [None] * world, []
            th = [threading.Thread(target=rank_main, args=(sw, r, prob, pset, instances, seed, out, err)) for r in range(world)]
            t0 = time.time()
            [t.start() for t in th]
            [t.join() for t in th]
            if err:
                print("nodes/step %d every %d ramp %d feed %d budget %d us world %d: FAILED %r" % (pset + (world, err[0])), flush=True)
                continue
            T = max(sw.clock)
            iters, nodes = out[0]["iters"], out[0]["nodes"]
            rate = iters / T
            if world == 1:
                base[pset] = rate
            b = base.get(pset) or base.get(sets[0]) or rate
            idle = max(o["idle_steps"] / max(1, o["steps"]) for o in out)
            print("nodes/step %2d exchange every %d ramp %d feed %2d budget %4d us | W %d: %5d nodes %8d it  %7.1f ms  %7.0f it/s  efficiency %.2f  busy %.2f"
                  "  worst rank idle %.0f %% of its steps  leaves moved %d  collectives %d  (%.0f s of CPU)"
                  % (pset + (world, nodes, iters, 1e3 * T, rate, rate / (world * b), sum(sw.busy) / (world * T), 100 * idle,
                             sum(o["moved"] for o in out), sw.colls, time.time() - t0)), flush=True)


if __name__ == "__main__":
    main()
