"""Hodgkin-Huxley squid-axon membrane model with Na/K leak, Na/K-ATPase pump and an optional
exponentially decaying synaptic stimulus -- the model of the reference's idealized-geometry
examples (reference: examples/idealized-geometries/mm_hh.py:7-161, mm_hh_no_stim.py), written
for the vectorised protocol: `rhs(t, states[n,4], parameters[n,17]) -> d states/dt [n,4]`,
with the channel currents I_ch_Na / I_ch_K / I_ch_Cl written into the parameter table as the
reference does (mm_hh.py:154-159).  SI units (V, s, S/m^2, mol/m^3)."""
import numpy as np

STATE_IND = dict(m=0, h=1, n=2, V=3)
PARAM_IND = dict(g_Na_bar=0, g_K_bar=1, g_leak_Na=2, g_leak_K=3, E_Na=4, E_K=5, Cm=6, stim_amplitude=7,
                 I_ch_Na=8, I_ch_K=9, I_ch_Cl=10, K_e=11, Na_i=12, m_K=13, m_Na=14, I_max=15, E_Cl=16)


def init_state_values(**values):
    init = np.array([0.016648440745822956, 0.8542015627820805, 0.1882020248041632,
                     -0.07438609374462003], dtype=np.float64)          # m, h, n, V  (mm_hh.py:12-15)
    for name, value in values.items():
        if name not in STATE_IND:
            raise ValueError("{0} is not a state.".format(name))
        init[STATE_IND[name]] = value
    return init


def init_parameter_values(**values):
    init = np.zeros(17, dtype=np.float64)
    init[[0, 1, 2, 3]] = [1200.0, 360.0, 2.0 * 0.5, 8.0 * 0.5]        # conductances (mm_hh.py:38-41)
    init[[13, 14, 15]] = [2.0, 7.7, 0.449]                             # pump m_K, m_Na, I_max (mm_hh.py:43-45)
    for name, value in values.items():
        if name not in PARAM_IND:
            raise ValueError("{0} is not a parameter.".format(name))
        init[PARAM_IND[name]] = value
    return init


def _indices(table, what, names):
    out = []
    for n in names:
        if n not in table:
            raise ValueError("Unknown {0}: '{1}'".format(what, n))
        out.append(table[n])
    return out if len(out) > 1 else out[0]


def state_indices(*states):
    return _indices(STATE_IND, "state", states)


def parameter_indices(*params):
    return _indices(PARAM_IND, "param", params)


def rhs_impl(t, states, parameters, with_stimulus):
    m, h, n, V = states[:, 0], states[:, 1], states[:, 2], states[:, 3]
    p = parameters
    u = 1.0e3 * (V + 65.0e-3)                                         # mV relative to -65 mV
    values = np.empty_like(states)
    alpha_m = 0.1e3 * (25.0 - u) / (np.exp((25.0 - u) / 10.0) - 1.0)
    beta_m = 4.0e3 * np.exp(-u / 18.0)
    values[:, 0] = (1 - m) * alpha_m - m * beta_m
    alpha_h = 0.07e3 * np.exp(-u / 20.0)
    beta_h = 1.0e3 / (np.exp((30.0 - u) / 10.0) + 1.0)
    values[:, 1] = (1 - h) * alpha_h - h * beta_h
    alpha_n = 0.01e3 * (10.0 - u) / (np.exp((10.0 - u) / 10.0) - 1.0)
    beta_n = 0.125e3 * np.exp(-u / 80.0)
    values[:, 2] = (1 - n) * alpha_n - n * beta_n
    i_pump = p[:, 15] / ((1 + p[:, 13] / p[:, 11]) ** 2 * (1 + p[:, 14] / p[:, 12]) ** 3)
    g_stim = p[:, 7] * np.exp(-np.mod(t, 0.03) / 0.002) * (t < 125e-3) if with_stimulus else 0.0
    i_Na = (p[:, 2] + p[:, 0] * h * m ** 3 + g_stim) * (V - p[:, 4]) + 3 * i_pump
    i_K = (p[:, 3] + p[:, 1] * n ** 4) * (V - p[:, 5]) - 2 * i_pump
    p[:, 8] = i_Na
    p[:, 9] = i_K
    p[:, 10] = 0.0
    values[:, 3] = (-i_K - i_Na) / p[:, 6]
    return values
