"""Experiment: two streaming pools (two engines) on ONE GPU in one process, one tree shared through dist.ShardedStream
with an in-process communicator and two host threads.  python two_pools.py [columns] [steps]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from miosqp_amd import problems, bnb, dist
from thread_comm import ThreadWorld, ThreadComm
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
W = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
world = ThreadWorld(W)
out = [None] * W
def body(rank):
    comm = ThreadComm(world, rank)
    st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
    m = bnb.MIOSQP()
    m.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st, dict(problems.QP_SETTINGS, max_batch=cols))
    sh = dist.ShardedStream(m, comm, columns=cols, exchange_every=4)
    eng = m.work.solver
    rng = np.random.RandomState(12345)
    def reroot():
        m.update_vectors(q=rng.randn(500), l=-2 + rng.rand(1000), u=2 + rng.rand(1000))
        sh.begin_instance()
    for k in range(steps):
        if sh.step() == 0:
            reroot()
        if k == steps // 2:
            comm.barrier()
            eng.batch_stats(reset=True); n0, i0, t1 = sh.ss.nodes, sh.ss.iters, time.perf_counter()
    comm.barrier()
    dt = time.perf_counter() - t1
    ms, lock, useful = eng.batch_stats()
    out[rank] = (sh.ss.iters - i0, sh.ss.nodes - n0, dt, ms, lock, useful, sh.moved)
ths = [threading.Thread(target=body, args=(r,)) for r in range(W)]
[t.start() for t in ths]; [t.join() for t in ths]
it = sum(o[0] for o in out); nd = sum(o[1] for o in out); dt = max(o[2] for o in out)
print("%d pools x %d columns: %.2f M node-it/s, %.0f nodes/s end to end; per pool device %.1f us per lock-step iteration, occupancy %s, moved %s"
      % (W, cols, it / dt * 1e-6, nd / dt, 1e3 * out[0][3] / max(1, out[0][4]), [round(o[5] / float(max(1, cols * o[4])), 3) for o in out], [o[6] for o in out]))
