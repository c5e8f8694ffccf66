"""Extended membrane ODE system of the EMIx calibration run (reference: examples/emix-simulations/mm_calibration.py:12-255): the
neuronal Hodgkin-Huxley membrane of mm_hh.py and the glial membrane of mm_glial.py coupled through compartment concentrations
(ECS, neuron, glia) that the channel currents change.  Integrated alone (no PDEs) until it is stationary, it yields the initial
membrane potentials, gating variables and concentrations of the full KNP-EMI run (run_calibration.py:13-90).  cm / ms / mV.
11 states, 12 parameters; vectorised protocol `rhs(t, states[n,11], parameters[n,12])`."""
import numpy as np

from knpemidg.models._hh_core import _indices

MODEL_ID = 6   # device model id of the batched HIP integrator (csrc/ode.hip)

STATE_IND = dict(m=0, h=1, n=2, V_n=3, V_g=4, K_e=5, K_n=6, K_g=7, Na_e=8, Na_n=9, Na_g=10)
PARAM_IND = dict(g_Na_bar=0, g_K_bar=1, g_leak_Na_n=2, g_leak_K_n=3, g_leak_Na_g=4, g_leak_K_g=5, Cm=6, stim_amplitude=7,
                 m_K=8, m_Na=9, I_max_n=10, I_max_g=11)

_TEMPERATURE, _R, _F = 300e3, 8.314e3, 96485e3            # mK, mJ/(K mol), mC/mol
_ICS_VOL, _ECS_VOL, _SURFACE = 3.42e-11 / 2.0, 7.08e-11, 2.29e-6          # cm^3, cm^3, cm^2 (mm_calibration.py:150-152)
_K_G_INIT, _K_E_INIT = 102.74050220804774, 3.32597273958481               # reference point of the Kir conductance (:154-155)


def init_state_values(**values):
    init = np.array([0.01, 0.85, 0.18, -74.38, -83.08, 3.32, 124.15, 102.75, 100.71, 12.83, 12.39])      # mm_calibration.py:19-33
    for name, value in values.items():
        if name not in STATE_IND:
            raise ValueError("{0} is not a state.".format(name))
        init[STATE_IND[name]] = value
    return init


def init_parameter_values(**values):
    init = np.array([120.0, 36.0, 0.1, 0.4, 0.1, 1.7, 2.0, 0.0, 2.0, 7.7, 44.9, 50.0])                   # mm_calibration.py:55-75
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
    s, p = states, parameters
    m, h, n, Vn, Vg = s[:, 0], s[:, 1], s[:, 2], s[:, 3], s[:, 4]
    K_e, K_n, K_g, Na_e, Na_n, Na_g = (s[:, k] for k in range(5, 11))
    c = _R * _TEMPERATURE / _F
    E_Na_n, E_K_n = c * np.log(Na_e / Na_n), c * np.log(K_e / K_n)
    E_Na_g, E_K_g = c * np.log(Na_e / Na_g), c * np.log(K_e / K_g)
    E_K_init = c * np.log(_K_E_INIT / _K_G_INIT)
    out = np.empty_like(s)
    alpha_m = 0.1 * (Vn + 40.0) / (1.0 - np.exp(-(Vn + 40.0) / 10.0))
    beta_m = 4.0 * np.exp(-(Vn + 65.0) / 18.0)
    alpha_h = 0.07 * np.exp(-(Vn + 65.0) / 20.0)
    beta_h = 1.0 / (1.0 + np.exp(-(Vn + 35.0) / 10.0))
    alpha_n = 0.01 * (Vn + 55.0) / (1.0 - np.exp(-(Vn + 55.0) / 10.0))
    beta_n = 0.125 * np.exp(-(Vn + 65.0) / 80.0)
    out[:, 0] = (1 - m) * alpha_m - m * beta_m
    out[:, 1] = (1 - h) * alpha_h - h * beta_h
    out[:, 2] = (1 - n) * alpha_n - n * beta_n
    g_stim = p[:, 7] * np.exp(-np.mod(t, 20.0) / 2.0)
    i_pump_n = p[:, 10] / ((1 + p[:, 8] / K_e) ** 2 * (1 + p[:, 9] / Na_n) ** 3)
    i_pump_g = p[:, 11] / ((1 + p[:, 8] / K_e) ** 2 * (1 + p[:, 9] / Na_g) ** 3)
    A = 1 + np.exp(18.4 / 42.4)
    B = 1 + np.exp(-(0.1186e3 + E_K_init) / 0.0441e3)
    C = 1 + np.exp((Vg - E_K_g + 0.0185e3) / 0.0425e3)
    D = 1 + np.exp(-(0.1186e3 + Vg) / 0.0441e3)
    i_Kir = p[:, 5] * np.sqrt(K_e / _K_E_INIT) * (A * B) / (C * D) * (Vg - E_K_g)
    i_Na_n = (p[:, 2] + p[:, 0] * h * m ** 3 + g_stim) * (Vn - E_Na_n) + 3 * i_pump_n
    i_K_n = (p[:, 3] + p[:, 1] * n ** 4) * (Vn - E_K_n) - 2 * i_pump_n
    i_Na_g = p[:, 4] * (Vg - E_Na_g) + 3 * i_pump_g
    i_K_g = i_Kir - 2 * i_pump_g
    out[:, 3] = (-i_K_n - i_Na_n) / p[:, 6]
    out[:, 4] = (-i_K_g - i_Na_g) / p[:, 6]
    ke, ki = _SURFACE / (_F * _ECS_VOL), _SURFACE / (_F * _ICS_VOL)
    out[:, 5] = (i_K_n + i_K_g) * ke
    out[:, 6] = -i_K_n * ki
    out[:, 7] = -i_K_g * ki
    out[:, 8] = (i_Na_n + i_Na_g) * ke
    out[:, 9] = -i_Na_n * ki
    out[:, 10] = -i_Na_g * ki
    return out
