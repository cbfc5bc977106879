"""Streaming batch on config 2: useful node-iterations per second, device occupancy.  python stream_rate.py [cols] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, bnb, stream
cols = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 1
pr = problems.random_miqp(**problems.CONFIGS["cfg2"], seed=0)
st = dict(problems.BNB_SETTINGS, max_iter_bb=10 ** 9)
model = bnb.MIOSQP()
model.setup(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"], pr["i_idx"], pr["i_l"], pr["i_u"], st, dict(problems.QP_SETTINGS, max_batch=cols))
margin = int(os.environ.get('STREAM_MARGIN', '0')) or None
srch = stream.StreamSearch(model, columns=cols, ring_margin=margin)
eng = model.work.solver
rng = np.random.RandomState(12345)
def reroot():
    model.update_vectors(q=rng.randn(500), l=-2 + rng.rand(1000), u=2 + rng.rand(1000))
    srch.begin_instance()
# ramp-up
t0 = time.perf_counter()
for k in range(steps):
    if srch.step(chunks) == 0:
        reroot()
    if k % 100 == 0:
        print(" step %d: nodes %d open %d in_flight %d free %d upper %.4f dropped %d" % (k, srch.nodes, len(srch.open), srch.in_flight, len(srch.free), model.work.upper_glob, srch.dropped))
    if k == steps // 2:
        eng.batch_stats(reset=True); n0, i0, t1 = srch.nodes, srch.iters, time.perf_counter()
dt = time.perf_counter() - t1
ms, lock, useful = eng.batch_stats()
print("cols %d: second half: %d nodes, %d iters in %.3f s -> %.2f M node-it/s end to end, %.0f nodes/s; device %.3f s (%.1f us per lock-step iteration), useful/(cols x lock-step) = %.3f, e2e/device time %.3f, dropped %d, open %d, free %d" % (
    cols, srch.nodes - n0, srch.iters - i0, dt, (srch.iters - i0) / dt * 1e-6, (srch.nodes - n0) / dt, ms * 1e-3, 1e3 * ms / max(1, lock),
    useful / float(max(1, cols * lock)), ms * 1e-3 / dt, srch.dropped, len(srch.open), len(srch.free)))
