"""cProfile of the power-converter MPC replay (config 4) on the HIP engine: where a step's ~0.7 ms goes."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from miosqp_amd import problems, qp  # noqa: E402

pc = problems.load_power_converter()
recs, model = problems.run_power_converter(pc, qp)
t0 = time.perf_counter()
recs, model = problems.run_power_converter(pc, qp, model=model)
dt = time.perf_counter() - t0
print("%d steps: %.1f us per step; solve run_time avg %.1f us" % (len(recs), 1e6 * dt / len(recs),
      1e6 * sum(r.get("run_time", 0.0) for r in recs) / len(recs)))
p = cProfile.Profile()
p.enable()
for _ in range(3):
    recs, model = problems.run_power_converter(pc, qp, model=model)
p.disable()
pstats.Stats(p).sort_stats("tottime").print_stats(18)
