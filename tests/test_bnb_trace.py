"""Host branch-and-bound control vs traces recorded from the reference (CPU, oracle backend).

With the same relaxation solver underneath (the CPU oracle), miosqp_amd.bnb must visit the same
nodes in the same order and report the same numbers as /root/reference/miosqp did when
tests/golden/make_bnb_traces.py recorded these fixtures.  Exact equality is required: both runs
go through the same update -> warm_start -> solve sequence (node.py:102-108).
"""
import numpy as np
import pytest

from golden_cases import case_names, load_case, run_case


@pytest.mark.parametrize("name", case_names())
def test_trace_matches_reference(name, oracle_mod):
    case = load_case(name)
    got = run_case(case, oracle_mod)
    assert len(got) == len(case["solves"])
    for g, e in zip(got, case["solves"]):
        assert g["status"] == e["status"]
        assert g["iter_num"] == e["iter_num"]
        assert g["osqp_iter"] == e["osqp_iter"]
        assert g["osqp_iter_avg"] == e["osqp_iter_avg"]
        assert g["trace"].shape == e["trace"].shape
        np.testing.assert_array_equal(g["trace"], e["trace"])
        assert g["upper_glob"] == e["upper_glob"]
        if e["status"] in ("Solved", "Max-iter feasible"):
            np.testing.assert_array_equal(g["x"], e["x"])


def test_golden_present():
    assert len(case_names()) >= 12
