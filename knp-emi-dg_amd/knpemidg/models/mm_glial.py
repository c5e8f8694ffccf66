"""Passive glial membrane with an inward-rectifying K channel (Kir 4.1), Na leak and Na/K-ATPase pump, cm / ms / mV
units (reference: examples/emix-simulations/mm_glial.py:6-170).  One state (the membrane potential), 19 parameters.
Vectorised protocol `rhs(t, states[n,1], parameters[n,19])`."""
import numpy as np

from knpemidg.models._hh_core import _indices

MODEL_ID = 4   # device model id of the batched HIP integrator (csrc/ode.hip)

STATE_IND = dict(V=0)
PARAM_IND = dict(g_Na_bar=0, g_K_bar=1, g_leak_Na=2, g_leak_K=3, E_Na=4, E_K=5, Cm=6, stim_amplitude=7,
                 I_ch_Na=8, I_ch_K=9, I_ch_Cl=10, K_e=11, Na_i=12, m_K=13, m_Na=14, I_max=15,
                 K_e_init=16, K_i_init=17, E_Cl=18)

_TEMPERATURE, _R, _F = 300e3, 8.314e3, 96485e3        # mK, mJ/(K mol), mC/mol


def init_state_values(**values):
    init = np.array([-83.08511451850003])
    for name, value in values.items():
        if name not in STATE_IND:
            raise ValueError("{0} is not a state.".format(name))
        init[STATE_IND[name]] = value
    return init


def init_parameter_values(**values):
    init = np.zeros(19, dtype=np.float64)
    init[[2, 3]] = [0.1, 1.7]                              # Na / K leak conductivities (mS/cm^2)
    init[[13, 14, 15]] = [2.0, 7.7, 50.0]                  # pump thresholds and strength
    init[[16, 17]] = [3.32597273958481, 102.74050220804774]
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
    i_pump = p[:, 15] / ((1 + p[:, 13] / p[:, 11]) ** 2 * (1 + p[:, 14] / p[:, 12]) ** 3)
    E_K_init = _R * _TEMPERATURE / _F * np.log(p[:, 16] / p[:, 17])
    dphi = V - p[:, 5]
    A = 1 + np.exp(18.4 / 42.4)
    B = 1 + np.exp(-(0.1186e3 + E_K_init) / 0.0441e3)
    C = 1 + np.exp((dphi + 0.0185e3) / 0.0425e3)
    D = 1 + np.exp(-(0.1186e3 + V) / 0.0441e3)
    g_Kir = np.sqrt(p[:, 11] / p[:, 16]) * (A * B) / (C * D)
    i_Kir = p[:, 3] * g_Kir * (V - p[:, 5])
    i_Na = p[:, 2] * (V - p[:, 4]) + 3 * i_pump
    i_K = i_Kir - 2 * i_pump
    p[:, 8] = i_Na
    p[:, 9] = i_K
    p[:, 10] = 0.0
    out = np.empty_like(states)
    out[:, 0] = (-i_K - i_Na) / p[:, 6]
    return out
