#!/usr/bin/env python
"""The reference's power-converter benchmark (horizon N = 3) on the MI355X engine.

The reference's script simulates the converter in closed loop and solves one MIQP per sampling step
(/root/reference/examples/power_converter/run_example.py:92-134, power_converter.py:589-675).  Here the MIQP
sequence of a recorded closed-loop run of the reference (tests/golden/power_converter_N3_long.npz: one settling
period + one measured period, 1600 steps) is replayed through MIOSQP.update_vectors / set_x0 / solve on the engine
under test; the inputs it applies are checked against the recorded ones step by step (same inputs = same plant
trajectory), then the reference's statistics are computed: solve-time columns of power_converter_timings.csv,
switching frequency and current THD.

    python examples/power_converter.py [--out results/power_converter_timings.csv] [--rho-auto]

(The same replay on the CPU restatement, for side-by-side numbers, is run by tests/side_by_side.py, which hands `run`
another backend module.)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from miosqp_amd import harness, problems  # noqa: E402


def run(backend=None, steps=None, rho_auto=False):
    """backend: a module with the osqp surface (None: miosqp_amd.qp, the HIP engine).  rho_auto: the sequence the reference
    recorded with rho chosen once at set-up."""
    if backend is None:
        from miosqp_amd import qp as backend
    pc = problems.load_power_converter(os.path.join(ROOT, "tests", "golden", "power_converter_N3_long_rhoauto.npz" if rho_auto
                                                    else "power_converter_N3_long.npz"))
    recs, _ = problems.run_power_converter(pc, backend, steps)
    U = np.array([r["x"][:6] for r in recs]).T
    worst = float(np.max(np.abs(U - pc["U"][:, :U.shape[1]])))
    first_timed = int(pc["init_periods"] * pc["Nstpp"])
    row = harness.timing_row(3, recs, min(first_timed, len(recs) - 1))
    if U.shape[1] == pc["U"].shape[1]:  # the whole run: the statistics of the loop this engine closed
        loop = dict(pc, U=U)
        row["fsw"], row["thd"] = harness.closed_loop_statistics(loop)
    row["max_input_deviation"] = worst
    row["nodes_per_step"] = float(np.mean([r["nodes"] for r in recs]))
    return row, pc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "results", "power_converter_timings.csv"))
    ap.add_argument("--rho-auto", action="store_true", help="the sequence recorded with rho chosen once at set-up")
    args = ap.parse_args()
    row, pc = run(rho_auto=args.rho_auto)
    cols = ["T", "miosqp_avg", "miosqp_std", "miosqp_min", "miosqp_max", "miosqp_osqp_avg_time",
            "miosqp_avg_osqp_iter", "fsw", "thd"]
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(",".join(cols) + "\n")
        f.write(",".join("%.6g" % row[c] for c in cols) + "\n")
    print("N = 3: solve time avg %.3f ms (std %.3f, min %.3f, max %.3f), relaxations %.1f %% of it, %.1f iterations/node, "
          "%.2f nodes/step" % (1e3 * row["miosqp_avg"], 1e3 * row["miosqp_std"], 1e3 * row["miosqp_min"],
                               1e3 * row["miosqp_max"], row["miosqp_osqp_avg_time"], row["miosqp_avg_osqp_iter"],
                               row["nodes_per_step"]))
    print("fsw %.3f Hz (reference run: %.3f), THD %.4f %% (reference run: %.4f), inputs differ from the recorded ones by "
          "at most %.1e" % (row["fsw"], pc["fsw"], row["thd"], pc["thd"], row["max_input_deviation"]))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
