"""Benchmark-harness parity: the statistics the reference's scripts compute, restated in
miosqp_amd/harness.py, against values computed by the reference itself (tests/golden/make_power_converter_long.py),
and the CSV schemas of examples/."""
import os
import subprocess
import sys

import numpy as np
import pytest

from miosqp_amd import harness, problems

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LONG = os.path.join(ROOT, "tests", "golden", "power_converter_N3_long.npz")
LONG_AUTO = os.path.join(ROOT, "tests", "golden", "power_converter_N3_long_rhoauto.npz")  # rho chosen once at set-up


@pytest.mark.parametrize("path", [LONG, LONG_AUTO])
def test_switching_frequency_and_thd_match_the_reference(path):
    pc = problems.load_power_converter(path)
    fsw, thd = harness.closed_loop_statistics(pc)
    assert abs(fsw - pc["fsw"]) <= 1e-12 * max(1.0, abs(pc["fsw"]))
    assert abs(thd - pc["thd"]) <= 1e-10 * max(1.0, abs(pc["thd"]))
    # the counting rule itself: 0 -> 1 turns switch 0 of a phase on, 1 -> 0 switch 2, 0 -> -1 switch 3, -1 -> 0 switch 1
    t = harness.on_transitions(np.array([1, 0, -1]), np.array([0, 1, 0]))
    assert list(np.nonzero(t)[0]) == [0, 6, 11]
    assert not harness.on_transitions(np.array([1, -1, 0]), np.array([-1, 1, 0])).any()  # two-level jumps do not count


@pytest.mark.parametrize("path", [LONG_AUTO])
def test_recorded_loop_with_rho_chosen_at_setup(oracle_mod, path):
    """the closed loop the reference recorded with `rho="auto"` in the shim (1600 MIQPs on ONE factor whose rho was chosen
    from the first step's vectors): the first 400 steps replayed on the oracle, count for count"""
    pc = problems.load_power_converter(path)
    assert pc["qp_settings"]["rho"] == "auto"
    recs, _ = problems.run_power_converter(pc, oracle_mod, 400)
    for k, r in enumerate(recs):
        assert r["status"] == "Solved"
        assert (r["nodes"], r["osqp_iter"]) == (int(pc["nodes"][k]), int(pc["osqp_iter"][k])), k
        np.testing.assert_array_equal(r["x"][:6], pc["U"][:, k])


@pytest.mark.gpu
def test_recorded_loop_with_rho_chosen_at_setup_on_gpu():
    """... and all 1600 steps on the HIP engine: the same rho, the same node and iteration counts, the recorded inputs"""
    from miosqp_amd import qp
    pc = problems.load_power_converter(LONG_AUTO)
    recs, model = problems.run_power_converter(pc, qp)
    assert len(recs) == 1600
    for k, r in enumerate(recs):
        assert r["status"] == "Solved"
        assert (r["nodes"], r["osqp_iter"]) == (int(pc["nodes"][k]), int(pc["osqp_iter"][k])), k
        assert np.max(np.abs(r["x"][:6] - pc["U"][:, k])) <= 1e-6


def test_power_converter_harness_closes_the_recorded_loop(oracle_mod, tmp_path):
    """The MIQP sequence replayed on the CPU oracle applies exactly the recorded inputs (first 400 steps here; the
    full run on the GPU is test_power_converter_harness_on_gpu) with the recorded node and iteration counts."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import power_converter as pcx
    pc = problems.load_power_converter(LONG)
    recs, _ = problems.run_power_converter(pc, oracle_mod, 400)
    for k, r in enumerate(recs):
        assert r["status"] == "Solved"
        assert (r["nodes"], r["osqp_iter"]) == (int(pc["nodes"][k]), int(pc["osqp_iter"][k])), k
        np.testing.assert_array_equal(r["x"][:6], pc["U"][:, k])
    row = harness.timing_row(3, recs, 200)
    assert row["miosqp_min"] <= row["miosqp_avg"] <= row["miosqp_max"] and 0 < row["miosqp_osqp_avg_time"] <= 100
    row2, _ = pcx.run(oracle_mod, steps=50)
    assert row2["max_input_deviation"] == 0.0 and "fsw" not in row2


def test_random_miqp_csv_schema(tmp_path):
    """examples/random_miqp.py writes the reference's columns (run_example.py:155-216, GUROBI columns aside)."""
    out = str(tmp_path / "grid.csv")
    env = dict(os.environ, PYTHONPATH=ROOT)
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "side_by_side.py"), "random_miqp",
                           "--repeat", "1", "--out", out], env=env, cwd=ROOT, timeout=600,
                          stdout=subprocess.DEVNULL)
    lines = open(out).read().strip().splitlines()
    assert lines[0] == "n,m,p,t_miosqp_avg,t_miosqp_std,t_miosqp_max,t_miosqp_osqp_avg,osqp_iter_avg"
    assert len(lines) == 9
    rows = [list(map(float, ln.split(","))) for ln in lines[1:]]
    assert [int(r[0]) for r in rows] == [10, 10, 50, 50, 100, 100, 150, 150]
    assert all(r[3] > 0 and 0 < r[6] <= 100 and r[7] > 0 for r in rows)


@pytest.mark.gpu
def test_power_converter_harness_on_gpu(tmp_path):
    """All 1600 steps on the HIP engine: the loop closes on the recorded inputs, so fsw and THD are the reference's."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import power_converter as pcx
    row, pc = pcx.run()
    assert row["max_input_deviation"] <= 1e-6
    assert abs(row["fsw"] - pc["fsw"]) <= 1e-9 and abs(row["thd"] - pc["thd"]) <= 1e-6
