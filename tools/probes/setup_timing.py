"""Set-up time of the engine at config 2, plain and with rho chosen at set-up (MIOSQP_SETUP_TIMING=1 prints the stages).
usage: setup_timing.py [config] [setup_on_device 0/1]"""
import os, sys, time
os.environ.setdefault("MIOSQP_SETUP_TIMING", "1")
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from miosqp_amd import problems, qp
pr = problems.random_miqp(**problems.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"], seed=0)
A, l, u = problems.extended(pr)
extra = dict(setup_on_device=int(sys.argv[2])) if len(sys.argv) > 2 else {}
for rho in (0.1, "auto", 0.1, "auto"):
    t0 = time.perf_counter()
    g = qp.OSQP()
    g.setup(pr["P"], pr["q"], A, l, u, **dict(problems.QP_SETTINGS, rho=rho, **extra))
    g.set_integer_rows(pr["i_idx"], pr["A"].shape[0])
    dt = time.perf_counter() - t0
    print("== rho=%s: set-up %.3f s, rho %.4g" % (rho, dt, g.rho()), flush=True)
    sys.stderr.flush()
    g.close()
