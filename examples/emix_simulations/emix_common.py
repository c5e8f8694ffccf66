"""EMIx configuration: dense reconstruction of brain tissue with neurons and glial cells, cm / ms / mV units
(reference: examples/emix-simulations/run_EMIx_simulation.py:54-262).  The mesh is the reference's bundled
`volume_ncells_5_size_5000` (22 419 vertices, 121 617 tets, labels 1..6), read from its XDMF/HDF5 files with the
pure-Python reader `knpemidg.h5lite` (no h5py / dolfin here).  The reference also reads facet tags from `tags.h5`, which
is not part of the repository; they are re-derived from cell-label disagreement the way the rat-neuron example does
(reference: examples/rat-neuron/run_rat_neuron.py:187-201)."""
import os
import sys
from collections import namedtuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(os.path.dirname(HERE)), "knp-emi-dg_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

from knpemidg import Solver, Constant                                        # noqa: E402
from knpemidg.mesh import Mesh, MeshFunction                                 # noqa: E402
from knpemidg.utils import pcws_constant_project, plus, minus                # noqa: E402
from knpemidg.models import mm_glial, mm_hh_emix                             # noqa: E402
from knpemidg.h5lite import read_xdmf_mesh                                   # noqa: E402

MESH_XDMF = os.path.join(HERE, "meshes", "volume_ncells_5_size_5000", "mesh.xdmf")        # the reference's bundled input mesh (data)
# label -> subdomain: 1 = ECS -> 0; 2, 3 = neurons -> 2; 4, 5, 6 = glial cells -> 1   (run_EMIx_simulation.py:173-185)
LABEL_TO_SUBDOMAIN = {1: 0, 2: 2, 3: 2, 4: 1, 5: 1, 6: 1}


class SolverEMIx(Solver):
    def __init__(self, params, ion_list, degree_emi=1, degree_knp=1, mms=None, sf=1):
        Solver.__init__(self, params, ion_list, degree_emi=degree_emi, degree_knp=degree_knp, mms=None, sf=sf)

    def update_ode(self, ode_model):
        # extracellular K and intracellular Na traces at the membrane (run_EMIx_simulation.py:39-50)
        K_e = plus(self.c_prev_k.split()[0], self.n_g)
        ode_model.set_parameter('K_e', pcws_constant_project(K_e, self.Q))
        Na_i = minus(self.ion_list[-1]['c'], self.n_g)
        ode_model.set_parameter('Na_i', pcws_constant_project(Na_i, self.Q))


def load_mesh(path=MESH_XDMF, refine=0):
    """(mesh [cm], subdomains, surfaces): membrane facet tag 1 = glial, 2 = neuronal, 10 = exterior boundary.
    refine = n: n regular refinements (every tet -> 8, labels inherited) -- the same tissue at 8^n times the cells, for measurements
    of the unstructured kernels at a roofline-relevant size (bench.py --workload emix --refine 1)."""
    coords, cells, attrs = read_xdmf_mesh(path)
    label = np.asarray(attrs["label"]).astype(np.int64)
    mesh = Mesh(coords, cells)
    for _ in range(int(refine)):
        from knpemidg.mesh import refine_uniform
        mesh, parent = refine_uniform(mesh)
        label = label[parent]
    sub = np.vectorize(LABEL_TO_SUBDOMAIN.get)(label).astype(np.uint32)
    fc = mesh.facet_cells
    interior = fc[:, 1] >= 0
    tags = np.zeros(mesh.num_facets(), dtype=np.uint32)
    tags[~interior] = 10
    l0, l1 = label[fc[interior, 0]], label[fc[interior, 1]]
    s0, s1 = sub[fc[interior, 0]], sub[fc[interior, 1]]
    differ = l0 != l1                                       # any two distinct biological cells are separated by a membrane
    kind = np.where((s0 == 2) | (s1 == 2), 2, 1)            # a neuron on either side -> neuronal membrane model
    t = np.zeros(int(interior.sum()), dtype=np.uint32)
    t[differ] = kind[differ]
    tags[interior] = t
    mesh.coords *= 1e-7                                     # nm -> cm (run_EMIx_simulation.py:224)
    return mesh, MeshFunction(mesh, 3, sub), MeshFunction(mesh, 2, tags)


def physical_setup(dt=0.1):
    C_M = 2.0
    temperature = 300e3
    F = 96485e3
    R = 8.314e3
    D_Na, D_K, D_Cl = 1.33e-8, 1.96e-8, 2.03e-8
    psi = F / (R * temperature)
    C_phi = C_M / dt
    K_e, K_n, K_g = 3.3236967382613933, 124.15397583492471, 102.75563828644862
    Na_e, Na_n, Na_g = 100.71925900028181, 12.838513108606818, 12.39731187972181
    Cl_e, Cl_n, Cl_g = Na_e + K_e, Na_n + K_n, Na_g + K_g
    rho_sub = {0: Constant(0), 1: Constant(0), 2: Constant(0)}
    params = namedtuple('params', ('dt', 'n_steps_ODE', 'F', 'psi', 'C_phi', 'C_M', 'R', 'temperature', 'phi_M_init_type',
                                   'rho_sub'))(dt, 25, F, psi, C_phi, C_M, R, temperature, 'constant', rho_sub)

    def ion(name, z, D, ce, cg, cn):
        return {'c_init_sub': {0: Constant(ce), 1: Constant(cg), 2: Constant(cn)}, 'c_init_sub_type': 'constant',
                'bdry': Constant(0), 'z': z, 'name': name, 'D_sub': {0: Constant(D), 1: Constant(D), 2: Constant(D)},
                'f_source': Constant(0)}
    Na = ion('Na', 1.0, D_Na, Na_e, Na_g, Na_n)
    K = ion('K', 1.0, D_K, K_e, K_g, K_n)
    Cl = ion('Cl', -1.0, D_Cl, Cl_e, Cl_g, Cl_n)
    g_syn_bar = 5
    stim = namedtuple('membrane_params', ('g_syn_bar', 'stimulus', 'stimulus_locator'))(
        g_syn_bar, {'stim_amplitude': g_syn_bar}, lambda x: (x[0] < 3.0e-4))
    return params, [K, Cl, Na], stim


def solver_parameters(**extra):
    # the reference's nominal tolerances (run_EMIx_simulation.py:226-243).  No per-mesh factors: the EMI solve stops on a residual
    # target derived from the concentration accuracy wanted and the KNP solve on a max-norm-like density test, with the same constants
    # as on the idealized meshes (knpemidg/solver.py; round 2 shipped emi_rtol_scale = 1e-4 and knp_rtol_scale = 0.03 for this mesh)
    names = ('direct_emi', 'direct_knp', 'rtol_emi', 'rtol_knp', 'atol_emi', 'atol_knp', 'threshold_emi', 'threshold_knp')
    vals = (False, False, 1E-5, 1E-7, 1E-40, 2E-40, 0.9, 0.75)
    return namedtuple('solver_params', names + tuple(extra))(*(vals + tuple(extra.values())))


def make_solver(dt=0.1, degree=1, verbose=False, mesh_tuple=None, refine=0):
    from knpemidg import setup_worker
    setup_worker.prestart(2)           # the hierarchy helpers start importing now, while the mesh is being built
    from knpemidg import _abi
    _abi._stamp("make_solver: start")
    params, ion_list, stim = physical_setup(dt)
    mesh, subdomains, surfaces = mesh_tuple or load_mesh(refine=refine)
    _abi._stamp("make_solver: mesh loaded")
    S = SolverEMIx(params, ion_list, degree_emi=degree, degree_knp=degree)
    S.verbose = verbose
    S.setup_domain(mesh, subdomains, surfaces)
    S.setup_parameters()
    S.setup_FEM_spaces()
    S.setup_membrane_model(stim, {1: mm_glial, 2: mm_hh_emix})                           # run_EMIx_simulation.py:249
    _abi._stamp("make_solver: membrane models attached")
    return S


def make_distributed_solver(rank, world, local_rank, dist, dt=0.1, degree=1, mesh_tuple=None):
    """BASELINE configs[4] on `world` GPUs: recursive-coordinate-bisection partition of the reconstruction, one ghost layer,
    RCCL halo exchange + all-reduced Krylov reductions (knpemidg/partition.py)."""
    from knpemidg.partition import distribute_solver
    params, ion_list, stim = physical_setup(dt)
    return distribute_solver(lambda: SolverEMIx(params, ion_list, degree_emi=degree, degree_knp=degree), mesh_tuple or load_mesh(),
                             {1: mm_glial, 2: mm_hh_emix}, stim, rank, world, local_rank, dist, method="rcb")
