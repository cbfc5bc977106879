"""BASELINE config 4: power_converter MPC horizon N=3 (n=18, 45 rows, l = -inf rows, P with
eigenvalues down to 1e-19): 40 consecutive MIQPs sharing one factorisation through
update_vectors + set_x0, recorded from the reference (tests/golden/make_power_converter.py)."""
import numpy as np
import pytest

from golden_cases import load_power_converter, run_power_converter


def test_mpc_sequence_matches_reference_cpu(oracle_mod):
    pc = load_power_converter()
    got = run_power_converter(pc, oracle_mod)
    for k, g in enumerate(got):
        assert g["status"] == pc["status"][k] == "Solved"
        assert g["nodes"] == pc["nodes"][k] and g["osqp_iter"] == pc["osqp_iter"][k], k
        assert g["upper"] == pc["upper"][k]
        np.testing.assert_array_equal(g["x"], pc["x"][k])


@pytest.mark.gpu
def test_mpc_sequence_matches_reference_gpu():
    from miosqp_amd import qp
    pc = load_power_converter()
    got = run_power_converter(pc, qp)
    for k, g in enumerate(got):
        assert g["status"] == pc["status"][k]
        assert g["nodes"] == pc["nodes"][k] and g["osqp_iter"] == pc["osqp_iter"][k], k
        assert abs(g["upper"] - pc["upper"][k]) <= 1e-8 * max(1.0, abs(pc["upper"][k]))
        np.testing.assert_allclose(g["x"], pc["x"][k], rtol=0, atol=1e-8)
