"""MI355X-native relaxation engine for miOSQP-style branch and bound.

Public surface mirrors the reference package (/root/reference/miosqp/__init__.py:1-3):
MIOSQP, add_bounds and the MI_* status strings.
"""
from miosqp_amd.bnb import (MIOSQP, Results, add_bounds, MI_UNSOLVED, MI_SOLVED,  # noqa: F401
                            MI_PRIMAL_INFEASIBLE, MI_DUAL_INFEASIBLE, MI_MAX_ITER_FEASIBLE,
                            MI_MAX_ITER_UNSOLVED)
