"""Idealized-geometry KNP-EMI configurations (reference: examples/idealized-geometries/run_2D.py:60-207,
run_3D.py:60-206): physical parameters, ion list [K, Cl, Na] (Na eliminated), HH membrane models,
stimulus on x < 20 um, solver tolerances.  Used by the run scripts, bench.py and the tests."""
import os
import sys
from collections import namedtuple

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-dg_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

from knpemidg import Solver, Constant, make_mesh_2D, make_mesh_3D          # noqa: E402
from knpemidg.utils import pcws_constant_project, plus, minus               # noqa: E402
from knpemidg.models import mm_hh, mm_hh_no_stim                            # noqa: E402


class SolverIdealized(Solver):
    """Sub-class providing the model-specific ODE hook (run_3D.py:31-51)."""

    def __init__(self, params, ion_list, degree_emi=1, degree_knp=1, mms=None, sf=1):
        Solver.__init__(self, params, ion_list, degree_emi=degree_emi, degree_knp=degree_knp, mms=None, sf=sf)

    def update_ode(self, ode_model):
        # extracellular trace of K and intracellular trace of Na at the membrane (run_3D.py:44-49)
        K_e = plus(self.c_prev_k.split()[0], self.n_g)
        ode_model.set_parameter('K_e', pcws_constant_project(K_e, self.Q))
        Na_i = minus(self.ion_list[-1]['c'], self.n_g)
        ode_model.set_parameter('Na_i', pcws_constant_project(Na_i, self.Q))


def physical_setup(dt=1.0e-4):
    C_M = 0.02
    temperature = 300
    F = 96485
    R = 8.314
    D_Na, D_K, D_Cl = 1.33e-9, 1.96e-9, 2.03e-9
    psi = F / (R * temperature)
    C_phi = C_M / dt
    Na_i_init, Na_e_init = 12.838513108648856, 100.71925900027354
    K_i_init, K_e_init = 124.15397583491901, 3.3236967382705265
    Cl_e_init = Na_e_init + K_e_init
    Cl_i_init = Na_i_init + K_i_init
    phi_M_init = Constant(-0.07438609374462003)
    rho_sub = {0: Constant(0), 1: Constant(0), 2: Constant(0)}
    params = namedtuple('params', ('dt', 'n_steps_ODE', 'F', 'psi', 'phi_M_init', 'C_phi', 'C_M', 'R', 'temperature',
                                   'phi_M_init_type', 'rho_sub'))(dt, 25, F, psi, phi_M_init, C_phi, C_M, R, temperature,
                                                                  'constant', rho_sub)

    def ion(name, z, D, ci, ce):
        return {'c_init_sub': {1: Constant(ci), 0: Constant(ce)}, 'c_init_sub_type': 'constant',
                'bdry': Constant(0), 'z': z, 'name': name, 'D_sub': {1: Constant(D), 0: Constant(D)},
                'f_source': Constant(0)}
    Na = ion('Na', 1.0, D_Na, Na_i_init, Na_e_init)
    K = ion('K', 1.0, D_K, K_i_init, K_e_init)
    Cl = ion('Cl', -1.0, D_Cl, Cl_i_init, Cl_e_init)
    ion_list = [K, Cl, Na]          # last ion is eliminated (run_3D.py:140-142)
    g_syn_bar = 10
    stim_params = namedtuple('membrane_params', ('g_syn_bar', 'stimulus', 'stimulus_locator'))(
        g_syn_bar, {'stim_amplitude': g_syn_bar}, lambda x: (x[0] < 20.0e-6))
    return params, ion_list, stim_params


def solver_parameters(dim, resolution, **extra):
    names = ('direct_emi', 'direct_knp', 'resolution', 'rtol_emi', 'rtol_knp', 'atol_emi', 'atol_knp',
             'threshold_emi', 'threshold_knp')
    vals = (False, False, resolution, 1E-5, 1E-7, 1E-40, 2E-40 if dim == 3 else 1E-40,
            0.9 if dim == 3 else None, 0.75 if dim == 3 else None)                      # run_3D.py:170-190
    names = names + tuple(extra.keys())
    vals = vals + tuple(extra.values())
    return namedtuple('solver_params', names)(*vals)


def make_solver(dim=3, resolution=0, n_axons=4, degree=1, dt=1.0e-4, verbose=False, mesh_tuple=None):
    """Build a ready-to-run solver for the 2D / 3D idealized geometry."""
    from knpemidg import setup_worker
    setup_worker.prestart(2)           # the hierarchy helpers start importing now, while the mesh is being built
    from knpemidg import _abi
    _abi._stamp("make_solver: start")
    params, ion_list, stim_params = physical_setup(dt)
    if mesh_tuple is None:
        mesh_tuple = make_mesh_3D(resolution, n_axons=n_axons) if dim == 3 else make_mesh_2D(resolution)
    _abi._stamp("make_solver: mesh built")
    mesh, subdomains, surfaces = mesh_tuple
    if dim == 3:
        ode_models = {1: mm_hh, 2: mm_hh_no_stim} if n_axons > 1 else {1: mm_hh}          # run_3D.py:196
    else:
        ode_models = {1: mm_hh}                                                           # run_2D.py:197
    S = SolverIdealized(params, ion_list, degree_emi=degree, degree_knp=degree)
    S.verbose = verbose
    S.setup_domain(mesh, subdomains, surfaces)
    S.setup_parameters()
    S.setup_FEM_spaces()
    S.setup_membrane_model(stim_params, ode_models)
    _abi._stamp("make_solver: membrane models attached")
    return S
