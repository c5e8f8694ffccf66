"""Passive membrane of the rat-neuron example: Na / K leak, Na/K-ATPase pump and a decaying synaptic conductance that adds
to the Na leak, SI units (reference: examples/rat-neuron/mm_leak.py:6-133).  One state (the membrane potential),
15 parameters.  Vectorised protocol `rhs(t, states[n,1], parameters[n,15])`."""
import numpy as np

from knpemidg.models._hh_core import _indices

MODEL_ID = 5   # device model id of the batched HIP integrator (csrc/ode.hip)

STATE_IND = dict(V=0)
PARAM_IND = dict(g_leak_Na=0, g_leak_K=1, E_Na=2, E_K=3, Cm=4, stim_amplitude=5, I_ch_Na=6, I_ch_K=7, I_ch_Cl=8,
                 K_e=9, Na_i=10, m_K=11, m_Na=12, I_max=13, E_Cl=14)


def init_state_values(**values):
    init = np.array([-0.07438609374462003])                # V (mm_leak.py:10)
    for name, value in values.items():
        if name not in STATE_IND:
            raise ValueError("{0} is not a state.".format(name))
        init[STATE_IND[name]] = value
    return init


def init_parameter_values(**values):
    init = np.zeros(15, dtype=np.float64)
    init[[0, 1]] = [2.0 * 0.5, 8.0 * 0.5]                  # leak conductivities, S/m^2 (mm_leak.py:30-31)
    init[[11, 12, 13]] = [2.0, 7.7, 0.449]                 # pump thresholds (mol/m^3) and strength (A/m^2)
    for name, value in values.items():
        if name not in PARAM_IND:
            raise ValueError("{0} is not a parameter.".format(name))
        init[PARAM_IND[name]] = value
    return init


def state_indices(*states):
    return _indices(STATE_IND, "state", states)


def parameter_indices(*params):
    return _indices(PARAM_IND, "param", params)


def rhs(t, states, parameters):
    V = states[:, 0]
    p = parameters
    g_stim = p[:, 5] * np.exp(-np.mod(t, 0.03) / 0.002)
    i_pump = p[:, 13] / ((1 + p[:, 11] / p[:, 9]) ** 2 * (1 + p[:, 12] / p[:, 10]) ** 3)
    i_Na = (p[:, 0] + g_stim) * (V - p[:, 2]) + 3 * i_pump
    i_K = p[:, 1] * (V - p[:, 3]) - 2 * i_pump
    p[:, 6] = i_Na
    p[:, 7] = i_K
    p[:, 8] = 0.0
    out = np.empty_like(states)
    out[:, 0] = (-i_K - i_Na) / p[:, 4]
    return out
