"""HH membrane without stimulus (reference: examples/idealized-geometries/mm_hh_no_stim.py)."""
from knpemidg.models._hh_core import (init_state_values, init_parameter_values, state_indices,
                                      parameter_indices, rhs_impl)

MODEL_ID = 2


def rhs(t, states, parameters):
    return rhs_impl(t, states, parameters, False)
