"""CPU oracle of the membrane ODE step (TEST INFRASTRUCTURE -- imported only by tests/, tests/golden/ generators and
bench.py's cpu_baseline leg; never by the product).

Restates, per membrane facet, what the reference does in `MembraneModel.step_lsoda`
(reference: src/knpemidg/membrane.py:84-119): overwrite the stimulus parameters on the masked rows, then one
LSODA call per row over [t, t+dt] with rtol = 1e-8, atol = 0 (`.0e-10`, membrane.py:112), keeping the last state.
LSODA here is scipy's `solve_ivp(method="LSODA")` (the same ODEPACK algorithm numbalsoda wraps); the right-hand
sides restate the gotran-generated models of the reference:

    hh(stim=True)   examples/idealized-geometries/mm_hh.py:118-161
    hh(stim=False)  examples/idealized-geometries/mm_hh_no_stim.py:118-159
    hh_emix         examples/emix-simulations/mm_hh.py:118-161        (cm / ms / mV; configs[4] neurons)
    glial           examples/emix-simulations/mm_glial.py:117-170     (cm / ms / mV; configs[4] glia, Kir 4.1)
    leak            examples/rat-neuron/mm_leak.py:107-133            (SI; passive dendrite / soma membrane)
    calibration     examples/emix-simulations/mm_calibration.py:143-255 (cm / ms / mV; ODE-only system behind configs[4]'s ICs)

State / parameter column layouts are the reference's (mm_hh.py:7-72).  The channel currents I_ch_k are a side
effect of the reference's last right-hand-side evaluation (mm_hh.py:154-159), i.e. tolerance-level quantities;
the oracle evaluates them at the end state.  Parity unpinned: the reference holds no numbers for this step
(SURVEY.md section 8c); independent of the product's Dormand-Prince integrator by construction.
"""
import math

import numpy as np
from scipy.integrate import solve_ivp

# column layout, mm_hh.py:56-64
P_IDX = {"g_Na_bar": 0, "g_K_bar": 1, "g_leak_Na": 2, "g_leak_K": 3, "E_Na": 4, "E_K": 5, "Cm": 6, "stim_amplitude": 7,
         "I_ch_Na": 8, "I_ch_K": 9, "I_ch_Cl": 10, "K_e": 11, "Na_i": 12, "m_K": 13, "m_Na": 14, "I_max": 15, "E_Cl": 16}
S_IDX = {"m": 0, "h": 1, "n": 2, "V": 3}


def hh_init_states():
    """mm_hh.py:7-17"""
    return np.array([0.016648440745822956, 0.8542015627820805, 0.1882020248041632, -0.07438609374462003])


def hh_init_parameters():
    """mm_hh.py:32-53"""
    p = np.zeros(17)
    p[0], p[1], p[2], p[3] = 1200.0, 360.0, 2.0 * 0.5, 8.0 * 0.5
    p[13], p[14], p[15] = 2.0, 7.7, 0.449
    return p


def hh_rhs(t, y, p, stim=True):
    """Right-hand side and the three channel currents (mm_hh.py:118-161), scalar Python floats."""
    m, h, n, V = y
    u = 1.0e3 * (V + 65.0e-3)
    alpha_m = 0.1e3 * (25.0 - u) / (math.exp((25.0 - u) / 10.0) - 1.0)
    beta_m = 4.0e3 * math.exp(-u / 18.0)
    alpha_h = 0.07e3 * math.exp(-u / 20.0)
    beta_h = 1.0e3 / (math.exp((30.0 - u) / 10.0) + 1.0)
    alpha_n = 0.01e3 * (10.0 - u) / (math.exp((10.0 - u) / 10.0) - 1.0)
    beta_n = 0.125e3 * math.exp(-u / 80.0)
    i_stim = p[7] * math.exp(-math.fmod(t, 0.03) / 0.002) * (1.0 if t < 125e-3 else 0.0) if stim else 0.0
    i_pump = p[15] / ((1.0 + p[13] / p[11]) ** 2 * (1.0 + p[14] / p[12]) ** 3)
    i_Na = (p[2] + p[0] * h * m ** 3 + i_stim) * (V - p[4]) + 3.0 * i_pump
    i_K = (p[3] + p[1] * n ** 4) * (V - p[5]) - 2.0 * i_pump
    dy = [(1.0 - m) * alpha_m - m * beta_m, (1.0 - h) * alpha_h - h * beta_h, (1.0 - n) * alpha_n - n * beta_n,
          (-i_K - i_Na) / p[6]]
    return dy, (i_Na, i_K, 0.0)


def step_lsoda(states, params, t0, dt, stim=True, stimulus=None, stimulus_mask=None, rtol=1.0e-8, atol=0.0):
    """Advance every row of `states` [n, 4] from t0 to t0+dt with its own LSODA call (membrane.py:98-114).
    `stimulus` = {parameter name: value} imposed on rows where `stimulus_mask` is True before the call
    (membrane.py:102-104).  Updates `states` and the I_ch columns of `params` in place."""
    n = states.shape[0]
    if stimulus:
        mask = np.ones(n, dtype=bool) if stimulus_mask is None else np.asarray(stimulus_mask, dtype=bool)
        for key, value in stimulus.items():
            params[mask, P_IDX[key]] = value
    for row in range(n):
        p = params[row]
        sol = solve_ivp(lambda t, y: hh_rhs(t, y, p, stim)[0], (t0, t0 + dt), states[row], method="LSODA", rtol=rtol,
                        atol=max(atol, 1e-300))
        assert sol.success                                         # membrane.py:113
        states[row] = sol.y[:, -1]
        _, cur = hh_rhs(t0 + dt, states[row], p, stim)
        p[8], p[9], p[10] = cur
    return states


# ---------------------------------------------------------------------------------------------------------------------
# The other membrane models of the reference, restated from its text (round 3): scalar Python floats, written
# independently of the product's vectorised numpy modules (knpemidg/models/*) and of csrc/ode.hip.

# column layout shared by mm_hh.py (emix) with the idealized model: examples/emix-simulations/mm_hh.py:47-54
HH_EMIX_P_IDX = P_IDX
# examples/emix-simulations/mm_glial.py:52-59
GLIAL_P_IDX = {"g_Na_bar": 0, "g_K_bar": 1, "g_leak_Na": 2, "g_leak_K": 3, "E_Na": 4, "E_K": 5, "Cm": 6, "stim_amplitude": 7,
               "I_ch_Na": 8, "I_ch_K": 9, "I_ch_Cl": 10, "K_e": 11, "Na_i": 12, "m_K": 13, "m_Na": 14, "I_max": 15,
               "K_e_init": 16, "K_i_init": 17, "E_Cl": 18}
# examples/rat-neuron/mm_leak.py:44-50
LEAK_P_IDX = {"g_leak_Na": 0, "g_leak_K": 1, "E_Na": 2, "E_K": 3, "Cm": 4, "stim_amplitude": 5, "I_ch_Na": 6, "I_ch_K": 7,
              "I_ch_Cl": 8, "K_e": 9, "Na_i": 10, "m_K": 11, "m_Na": 12, "I_max": 13, "E_Cl": 14}


def hh_emix_init_states():
    """examples/emix-simulations/mm_hh.py:11-16 (m, h, n, V in mV)"""
    return np.array([0.016651023270342777, 0.8541791472445746, 0.18821645700362638, -74.3848784437955])


def hh_emix_init_parameters():
    """examples/emix-simulations/mm_hh.py:33-45 (mS/cm^2, uA/cm^2)"""
    p = np.zeros(17)
    p[0], p[1], p[2], p[3] = 120.0, 36.0, 0.1, 0.4
    p[13], p[14], p[15] = 2.0, 7.7, 44.9
    return p


def hh_emix_rhs(t, y, p):
    """examples/emix-simulations/mm_hh.py:118-161: HH kinetics in mV / ms, stimulus conductance re-triggered every 20 ms
    (the `*(t < 45)` factor is commented out in the reference)."""
    m, h, n, V = y
    alpha_m = 0.1 * (V + 40.0) / (1.0 - math.exp(-(V + 40.0) / 10.0))
    beta_m = 4.0 * math.exp(-(V + 65.0) / 18.0)
    alpha_h = 0.07 * math.exp(-(V + 65.0) / 20.0)
    beta_h = 1.0 / (1.0 + math.exp(-(V + 35.0) / 10.0))
    alpha_n = 0.01 * (V + 55.0) / (1.0 - math.exp(-(V + 55.0) / 10.0))
    beta_n = 0.125 * math.exp(-(V + 65) / 80.0)
    i_stim = p[7] * math.exp(-math.fmod(t, 20.0) / 2.0)
    i_pump = p[15] / ((1.0 + p[13] / p[11]) ** 2 * (1.0 + p[14] / p[12]) ** 3)
    i_Na = (p[2] + p[0] * h * math.pow(m, 3) + i_stim) * (V - p[4]) + 3.0 * i_pump
    i_K = (p[3] + p[1] * math.pow(n, 4)) * (V - p[5]) - 2.0 * i_pump
    dy = [(1.0 - m) * alpha_m - m * beta_m, (1.0 - h) * alpha_h - h * beta_h, (1.0 - n) * alpha_n - n * beta_n,
          (-i_K - i_Na) / p[6]]
    return dy, (i_Na, i_K, 0.0)


def glial_init_states():
    """examples/emix-simulations/mm_glial.py:11"""
    return np.array([-83.08511451850003])


def glial_init_parameters():
    """examples/emix-simulations/mm_glial.py:34-50"""
    p = np.zeros(19)
    p[2], p[3] = 0.1, 1.7
    p[13], p[14], p[15] = 2.0, 7.7, 50.0
    p[16], p[17] = 3.32597273958481, 102.74050220804774
    return p


def glial_rhs(t, y, p):
    """examples/emix-simulations/mm_glial.py:117-170: Na leak, Kir 4.1 with the conductance of Halnes et al. 2013 in the
    form the reference uses (square root in K_e / K_e_init, four Boltzmann factors), Na/K-ATPase 3 : 2."""
    V = y[0]
    i_pump = p[15] / ((1.0 + p[13] / p[11]) ** 2 * (1.0 + p[14] / p[12]) ** 3)
    temperature, R, F = 300e3, 8.314e3, 96485e3                            # mm_glial.py:139-141 (mK, mJ/(K mol), mC/mol)
    E_K_init = R * temperature / F * math.log(p[16] / p[17])
    dphi = V - p[5]
    A = 1.0 + math.exp(18.4 / 42.4)
    B = 1.0 + math.exp(-(0.1186e3 + E_K_init) / 0.0441e3)
    C = 1.0 + math.exp((dphi + 0.0185e3) / 0.0425e3)
    D = 1.0 + math.exp(-(0.1186e3 + V) / 0.0441e3)
    g_Kir = math.sqrt(p[11] / p[16]) * (A * B) / (C * D)
    i_Kir = p[3] * g_Kir * (V - p[5])
    i_Na = p[2] * (V - p[4]) + 3.0 * i_pump
    i_K = i_Kir - 2.0 * i_pump
    return [(-i_K - i_Na) / p[6]], (i_Na, i_K, 0.0)


def leak_init_states():
    """examples/rat-neuron/mm_leak.py:10"""
    return np.array([-0.07438609374462003])


def leak_init_parameters():
    """examples/rat-neuron/mm_leak.py:29-41"""
    p = np.zeros(15)
    p[0], p[1] = 2.0 * 0.5, 8.0 * 0.5
    p[11], p[12], p[13] = 2.0, 7.7, 0.449
    return p


def leak_rhs(t, y, p):
    """examples/rat-neuron/mm_leak.py:107-133: Na / K leak + pump; the synaptic conductance adds to the Na leak."""
    V = y[0]
    i_stim = p[5] * math.exp(-math.fmod(t, 0.03) / 0.002)
    i_pump = p[13] / ((1.0 + p[11] / p[9]) ** 2 * (1.0 + p[12] / p[10]) ** 3)
    i_Na = (p[0] + i_stim) * (V - p[2]) + 3.0 * i_pump
    i_K = p[1] * (V - p[3]) - 2.0 * i_pump
    return [(-i_K - i_Na) / p[4]], (i_Na, i_K, 0.0)


def calibration_init_states():
    """examples/emix-simulations/mm_calibration.py:19-33 (m h n V_n V_g K_e K_n K_g Na_e Na_n Na_g)"""
    return np.array([0.01, 0.85, 0.18, -74.38, -83.08, 3.32, 124.15, 102.75, 100.71, 12.83, 12.39])


def calibration_init_parameters():
    """examples/emix-simulations/mm_calibration.py:55-75"""
    return np.array([120.0, 36.0, 0.1, 0.4, 0.1, 1.7, 2.0, 0.0, 2.0, 7.7, 44.9, 50.0])


CALIBRATION_P_IDX = {"g_Na_bar": 0, "g_K_bar": 1, "g_leak_Na_n": 2, "g_leak_K_n": 3, "g_leak_Na_g": 4, "g_leak_K_g": 5, "Cm": 6,
                     "stim_amplitude": 7, "m_K": 8, "m_Na": 9, "I_max_n": 10, "I_max_g": 11}


def calibration_rhs(t, y, p):
    """examples/emix-simulations/mm_calibration.py:143-255: neuronal HH + glial Kir membranes with compartment concentrations as
    states (Nernst potentials follow them); returns (dy, (neuronal Na current, neuronal K current, glial K current))."""
    temperature, R, F = 300e3, 8.314e3, 96485e3
    ICS_vol, ECS_vol, surface = 3.42e-11 / 2.0, 7.08e-11, 2.29e-6
    K_g_init, K_e_init = 102.74050220804774, 3.32597273958481
    m, h, n, Vn, Vg, K_e, K_n, K_g, Na_e, Na_n, Na_g = y
    E_Na_n = R * temperature / F * math.log(Na_e / Na_n)
    E_K_n = R * temperature / F * math.log(K_e / K_n)
    E_Na_g = R * temperature / F * math.log(Na_e / Na_g)
    E_K_g = R * temperature / F * math.log(K_e / K_g)
    E_K_init = R * temperature / F * math.log(K_e_init / K_g_init)
    alpha_m = 0.1 * (Vn + 40.0) / (1.0 - math.exp(-(Vn + 40.0) / 10.0))
    beta_m = 4.0 * math.exp(-(Vn + 65.0) / 18.0)
    alpha_h = 0.07 * math.exp(-(Vn + 65.0) / 20.0)
    beta_h = 1.0 / (1.0 + math.exp(-(Vn + 35.0) / 10.0))
    alpha_n = 0.01 * (Vn + 55.0) / (1.0 - math.exp(-(Vn + 55.0) / 10.0))
    beta_n = 0.125 * math.exp(-(Vn + 65) / 80.0)
    i_Stim = p[7] * math.exp(-math.fmod(t, 20.0) / 2.0)
    i_pump_n = p[10] / ((1 + p[8] / K_e) ** 2 * (1 + p[9] / Na_n) ** 3)
    i_pump_g = p[11] / ((1 + p[8] / K_e) ** 2 * (1 + p[9] / Na_g) ** 3)
    dphi = Vg - E_K_g
    A = 1 + math.exp(18.4 / 42.4)
    B = 1 + math.exp(-(0.1186e3 + E_K_init) / 0.0441e3)
    C = 1 + math.exp((dphi + 0.0185e3) / 0.0425e3)
    D = 1 + math.exp(-(0.1186e3 + Vg) / 0.0441e3)
    g_Kir = math.sqrt(K_e / K_e_init) * (A * B) / (C * D)
    I_Kir = p[5] * g_Kir * (Vg - E_K_g)
    i_Na_n = (p[2] + p[0] * h * math.pow(m, 3) + i_Stim) * (Vn - E_Na_n) + 3 * i_pump_n
    i_K_n = (p[3] + p[1] * math.pow(n, 4)) * (Vn - E_K_n) - 2 * i_pump_n
    i_Na_g = p[4] * (Vg - E_Na_g) + 3 * i_pump_g
    i_K_g = I_Kir - 2 * i_pump_g
    dy = [(1 - m) * alpha_m - m * beta_m, (1 - h) * alpha_h - h * beta_h, (1 - n) * alpha_n - n * beta_n,
          (-i_K_n - i_Na_n) / p[6], (-i_K_g - i_Na_g) / p[6],
          i_K_n * surface / (F * ECS_vol) + i_K_g * surface / (F * ECS_vol), -i_K_n * surface / (F * ICS_vol),
          -i_K_g * surface / (F * ICS_vol), i_Na_n * surface / (F * ECS_vol) + i_Na_g * surface / (F * ECS_vol),
          -i_Na_n * surface / (F * ICS_vol), -i_Na_g * surface / (F * ICS_vol)]
    return dy, (i_Na_n, i_K_n, i_K_g)


def step_lsoda_plain(rhs, states, params, t0, dt, rtol=1.0e-8, atol=0.0):
    """One LSODA call per row for systems that keep no currents in their parameter table (the calibration system)."""
    for row in range(states.shape[0]):
        p = params[row]
        sol = solve_ivp(lambda t, y: rhs(t, y, p)[0], (t0, t0 + dt), states[row], method="LSODA", rtol=rtol, atol=max(atol, 1e-300))
        assert sol.success
        states[row] = sol.y[:, -1]
    return states


# name -> (initial states, initial parameters, rhs(t, y, p) -> (dy, currents), parameter index table, state index of V)
MODELS = {
    "hh": (hh_init_states, hh_init_parameters, lambda t, y, p: hh_rhs(t, y, p, True), P_IDX, 3),
    "hh_no_stim": (hh_init_states, hh_init_parameters, lambda t, y, p: hh_rhs(t, y, p, False), P_IDX, 3),
    "hh_emix": (hh_emix_init_states, hh_emix_init_parameters, hh_emix_rhs, HH_EMIX_P_IDX, 3),
    "glial": (glial_init_states, glial_init_parameters, glial_rhs, GLIAL_P_IDX, 0),
    "leak": (leak_init_states, leak_init_parameters, leak_rhs, LEAK_P_IDX, 0),
}


def step_lsoda_model(model, states, params, t0, dt, stimulus=None, stimulus_mask=None, rtol=1.0e-8, atol=0.0):
    """`step_lsoda` for any entry of MODELS: one LSODA call per row (membrane.py:98-114), stimulus parameters re-imposed on
    the masked rows first (membrane.py:102-104), channel currents evaluated at the end state."""
    _, _, rhs, pidx, _ = MODELS[model]
    n = states.shape[0]
    if stimulus:
        mask = np.ones(n, dtype=bool) if stimulus_mask is None else np.asarray(stimulus_mask, dtype=bool)
        for key, value in stimulus.items():
            params[mask, pidx[key]] = value
    for row in range(n):
        p = params[row]
        sol = solve_ivp(lambda t, y: rhs(t, y, p)[0], (t0, t0 + dt), states[row], method="LSODA", rtol=rtol,
                        atol=max(atol, 1e-300))
        assert sol.success                                         # membrane.py:113
        states[row] = sol.y[:, -1]
        _, cur = rhs(t0 + dt, states[row], p)
        p[pidx["I_ch_Na"]], p[pidx["I_ch_K"]], p[pidx["I_ch_Cl"]] = cur
    return states


class MembraneOracle:
    """ODE tables of one membrane tag + the PDE<->ODE copies of the reference's time loop
    (reference: src/knpemidg/solver.py:1076-1113; update_ode hook: examples/idealized-geometries/run_3D.py:39-51).
    `model` names an entry of MODELS (default: the idealized HH model, with / without stimulus by `stim`)."""

    def __init__(self, pb, tag, stim, C_M, model=None):
        self.tag = int(tag)
        self.stim = bool(stim)
        self.model = model if model is not None else ("hh" if stim else "hh_no_stim")
        init_s, init_p, _, self.pidx, self.iV = MODELS[self.model]
        self.fids = np.nonzero((pb.mesh.facet_cells[:, 1] >= 0) & (pb.facet_tags == self.tag))[0]
        n = len(self.fids)
        self.states = np.tile(init_s(), (n, 1))
        self.params = np.tile(init_p(), (n, 1))
        self.params[:, self.pidx["Cm"]] = C_M                                     # solver.py:248
        self.x = pb.mesh.facet_midpoints()[self.fids]
        self.time = 0.0


def oracle_membrane_step(pb, E, models, k, dt, stimulus, stimulus_locator, phi_M_init_constant=True):
    """One pass of solver.py:1076-1113 over all membrane models; writes pb.phi_M and pb.I_ch on their facets.
    E = {ion name: Nernst potential on pb.mem facets} from the previous PDE step."""
    import knpemi_oracle as ko
    pos = {f: i for i, f in enumerate(pb.mem)}
    for M in models:
        f = M.fids
        if not (phi_M_init_constant and k == 0):
            M.states[:, M.iV] = pb.phi_M[f]                                    # solver.py:1086-1094
        sel = np.array([pos[i] for i in f], dtype=np.int64)
        for ion in pb.ions:
            M.params[:, M.pidx["E_" + ion["name"]]] = E[ion["name"]][sel]      # solver.py:1097-1098
        # update_ode: K_e = facet-avg plus(c_K), Na_i = facet-avg minus(c_Na)   (run_3D.py:44-49,
        # examples/emix-simulations/run_EMIx_simulation.py: the same hook)
        names = [ion["name"] for ion in pb.ions]
        cc = pb.all_c()
        M.params[:, M.pidx["K_e"]] = ko.facet_average(pb, f, lambda plus, minus: plus(cc[names.index("K")]), max(1, pb.p))
        M.params[:, M.pidx["Na_i"]] = ko.facet_average(pb, f, lambda plus, minus: minus(cc[names.index("Na")]), max(1, pb.p))
        mask = np.fromiter(map(stimulus_locator, M.x), dtype=bool, count=len(f))
        step_lsoda_model(M.model, M.states, M.params, M.time, dt, stimulus=stimulus, stimulus_mask=mask)
        M.time += dt
        pb.phi_M[f] = M.states[:, M.iV]                                        # solver.py:1107
        for ion in pb.ions:
            pb.I_ch[ion["name"]][f] = M.params[:, M.pidx["I_ch_" + ion["name"]]]  # solver.py:1110-1112
