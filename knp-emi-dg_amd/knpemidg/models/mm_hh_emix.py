"""Hodgkin-Huxley neuron membrane of the EMIx simulations in cm / ms / mV units (reference:
examples/emix-simulations/mm_hh.py:7-161): same gating kinetics as the idealized-geometry model, synaptic
conductance exp(-mod(t, 20)/2), Na/K leak and Na/K-ATPase pump.  Vectorised protocol
`rhs(t, states[n,4], parameters[n,17])`."""
import numpy as np

from knpemidg.models._hh_core import _indices

MODEL_ID = 3   # device model id of the batched HIP integrator (csrc/ode.hip)

STATE_IND = dict(m=0, h=1, n=2, V=3)
PARAM_IND = dict(g_Na_bar=0, g_K_bar=1, g_leak_Na=2, g_leak_K=3, E_Na=4, E_K=5, Cm=6, stim_amplitude=7,
                 I_ch_Na=8, I_ch_K=9, I_ch_Cl=10, K_e=11, Na_i=12, m_K=13, m_Na=14, I_max=15, E_Cl=16)


def init_state_values(**values):
    init = np.array([0.016651023270342777, 0.8541791472445746, 0.18821645700362638, -74.3848784437955])   # m, h, n, V
    for name, value in values.items():
        if name not in STATE_IND:
            raise ValueError("{0} is not a state.".format(name))
        init[STATE_IND[name]] = value
    return init


def init_parameter_values(**values):
    init = np.zeros(17, dtype=np.float64)
    init[[0, 1, 2, 3]] = [120.0, 36.0, 0.1, 0.4]          # mS/cm^2
    init[[13, 14, 15]] = [2.0, 7.7, 44.9]                 # pump thresholds (mol/m^3) and strength (uA/cm^2)
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
    m, h, n, V = states[:, 0], states[:, 1], states[:, 2], states[:, 3]
    p = parameters
    out = np.empty_like(states)
    alpha_m = 0.1 * (V + 40.0) / (1.0 - np.exp(-(V + 40.0) / 10.0))
    beta_m = 4.0 * np.exp(-(V + 65.0) / 18.0)
    alpha_h = 0.07 * np.exp(-(V + 65.0) / 20.0)
    beta_h = 1.0 / (1.0 + np.exp(-(V + 35.0) / 10.0))
    alpha_n = 0.01 * (V + 55.0) / (1.0 - np.exp(-(V + 55.0) / 10.0))
    beta_n = 0.125 * np.exp(-(V + 65.0) / 80.0)
    out[:, 0] = (1 - m) * alpha_m - m * beta_m
    out[:, 1] = (1 - h) * alpha_h - h * beta_h
    out[:, 2] = (1 - n) * alpha_n - n * beta_n
    g_stim = p[:, 7] * np.exp(-np.mod(t, 20.0) / 2.0)
    i_pump = p[:, 15] / ((1 + p[:, 13] / p[:, 11]) ** 2 * (1 + p[:, 14] / p[:, 12]) ** 3)
    i_Na = (p[:, 2] + p[:, 0] * h * m ** 3 + g_stim) * (V - p[:, 4]) + 3 * i_pump
    i_K = (p[:, 3] + p[:, 1] * n ** 4) * (V - p[:, 5]) - 2 * i_pump
    p[:, 8] = i_Na
    p[:, 9] = i_K
    p[:, 10] = 0.0
    out[:, 3] = (-i_K - i_Na) / p[:, 6]
    return out
