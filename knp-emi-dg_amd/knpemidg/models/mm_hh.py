"""HH membrane with synaptic stimulus (reference: examples/idealized-geometries/mm_hh.py)."""
from knpemidg.models._hh_core import (init_state_values, init_parameter_values, state_indices,
                                      parameter_indices, rhs_impl)

MODEL_ID = 1   # device model id of the batched HIP integrator


def rhs(t, states, parameters):
    return rhs_impl(t, states, parameters, True)
