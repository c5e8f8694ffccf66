"""CPU oracle of the membrane ODE step (TEST INFRASTRUCTURE -- imported only by tests/, tests/golden/ generators and
bench.py's cpu_baseline leg; never by the product).

Restates, per membrane facet, what the reference does in `MembraneModel.step_lsoda`
(reference: src/knpemidg/membrane.py:84-119): overwrite the stimulus parameters on the masked rows, then one
LSODA call per row over [t, t+dt] with rtol = 1e-8, atol = 0 (`.0e-10`, membrane.py:112), keeping the last state.
LSODA here is scipy's `solve_ivp(method="LSODA")` (the same ODEPACK algorithm numbalsoda wraps); the right-hand
sides restate the gotran-generated models of the reference:

    hh(stim=True)   examples/idealized-geometries/mm_hh.py:118-161
    hh(stim=False)  examples/idealized-geometries/mm_hh_no_stim.py:118-159

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


class MembraneOracle:
    """ODE tables of one membrane tag + the PDE<->ODE copies of the reference's time loop
    (reference: src/knpemidg/solver.py:1076-1113; update_ode hook: examples/idealized-geometries/run_3D.py:39-51)."""

    def __init__(self, pb, tag, stim, C_M):
        self.tag = int(tag)
        self.stim = bool(stim)
        self.fids = np.nonzero((pb.mesh.facet_cells[:, 1] >= 0) & (pb.facet_tags == self.tag))[0]
        n = len(self.fids)
        self.states = np.tile(hh_init_states(), (n, 1))
        self.params = np.tile(hh_init_parameters(), (n, 1))
        self.params[:, P_IDX["Cm"]] = C_M                                     # solver.py:248
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
            M.states[:, S_IDX["V"]] = pb.phi_M[f]                              # solver.py:1086-1094
        sel = np.array([pos[i] for i in f], dtype=np.int64)
        for ion in pb.ions:
            M.params[:, P_IDX["E_" + ion["name"]]] = E[ion["name"]][sel]       # solver.py:1097-1098
        # update_ode: K_e = facet-avg plus(c_K), Na_i = facet-avg minus(c_Na)   (run_3D.py:44-49)
        names = [ion["name"] for ion in pb.ions]
        cc = pb.all_c()
        M.params[:, P_IDX["K_e"]] = ko.facet_average(pb, f, lambda plus, minus: plus(cc[names.index("K")]), max(1, pb.p))
        M.params[:, P_IDX["Na_i"]] = ko.facet_average(pb, f, lambda plus, minus: minus(cc[names.index("Na")]), max(1, pb.p))
        mask = np.fromiter(map(stimulus_locator, M.x), dtype=bool, count=len(f))
        step_lsoda(M.states, M.params, M.time, dt, stim=M.stim, stimulus=stimulus, stimulus_mask=mask)
        M.time += dt
        pb.phi_M[f] = M.states[:, S_IDX["V"]]                                  # solver.py:1107
        for ion in pb.ions:
            pb.I_ch[ion["name"]][f] = M.params[:, P_IDX["I_ch_" + ion["name"]]]  # solver.py:1110-1112
