"""Edge cases and error behaviour of the device path (failure detection; section 8b: errors)."""
import os
import sys

import numpy as np
import pytest

import knpemi_oracle as ko
from common import synthetic_state, device_for, push_state, relerr, small_3d

pytestmark = pytest.mark.gpu


def test_no_membrane_facets(hip_lib):
    """A mesh without any membrane tag: pure-Neumann EMI (still singular), KNP decoupled; no ODE nodes."""
    from knpemidg import _abi as A
    from knpemidg.mesh import BoxMesh, MeshFunction
    m = BoxMesh((0, 0, 0), (4e-6, 0.3e-6, 0.3e-6), 4, 3, 3)
    s = MeshFunction(m, 3, 0)
    f = MeshFunction(m, 2, 0)
    f.array()[m.exterior_facets()] = 5
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    assert len(pb.mem) == 0
    x = synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.update_dnphi()
    Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
    dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < 1e-11
    dev.emi_rhs(); dev.knp_rhs()
    assert relerr(dev.download(A.F_B_EMI), b) < 1e-11
    dev.step_updates()                                   # no membrane facets: must be a no-op for facet fields
    assert np.all(dev.download(A.F_PHI_M) == pb.phi_M)
    dev.close()


def test_two_cell_mesh_and_single_species(hip_lib):
    """Smallest possible meshes / species counts (ragged: nc far below one workgroup)."""
    from knpemidg import _abi as A
    from knpemidg.mesh import Mesh, MeshFunction
    coords = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1.0]]) * 1e-6
    m = Mesh(coords, np.array([[0, 1, 2, 3], [1, 2, 3, 4]]))
    s = MeshFunction(m, 3, np.array([0, 1]))
    f = MeshFunction(m, 2, 0)
    f.array()[m.interior_facets()] = 1
    for n_ions in (2, 3):
        P = ko.idealized_params()
        ions = [dict(name=n, z=P["z"][n], D=np.full(2, P["D"][n])) for n in ("K", "Cl", "Na")][3 - n_ions:]
        pb = ko.Problem(m, s.array(), f.array(), 1, ions, P, membrane_tags=(1,))
        rng = np.random.default_rng(5)
        pb.c = 50 + 10 * rng.uniform(size=pb.c.shape); pb.c_prev_n = pb.c.copy()
        pb.c_elim = 60 + 10 * rng.uniform(size=pb.c_elim.shape)
        pb.phi = 0.05 * rng.uniform(-1, 1, size=pb.phi.shape)
        pb.phi_M[pb.mem] = -0.07
        dev = device_for(pb)
        push_state(dev, pb)
        dev.update_kappa(); dev.update_dnphi()
        x = rng.uniform(-1, 1, size=(pb.N_ions, 2, 4))
        Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
        dev.upload(A.F_X, np.concatenate([x[0].ravel(), np.zeros((pb.N_ions - 1) * 8)])); dev.emi_apply(A.F_X, A.F_Y)
        assert relerr(dev.download(A.F_Y, 0, 8), Aemi @ x[0].ravel()) < 1e-11
        dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < 1e-11
        dev.knp_rhs()
        bk = dev.download(A.F_B_KNP).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(bk[k], ko.knp_rhs(pb, k)) < 1e-11
        dev.close()


def test_nonconvergence_raises(hip_lib):
    """maxit too small -> status -3 -> KnpError (ksp_error_if_not_converged, solver.py:428)."""
    from knpemidg import _abi as A
    m, s, f = small_3d()
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.emi_rhs()
    dev.upload(A.F_PHI, np.zeros(pb.ndof))
    with pytest.raises(A.KnpError, match="did not converge"):
        dev.emi_solve(1e-12, maxit=3, check_every=1)
    dev.update_dnphi(); dev.knp_rhs()
    with pytest.raises(A.KnpError, match="did not converge"):
        dev.knp_solve(1e-14, maxit=6, min_it=5, check_every=1)
    dev.close()


def test_bad_arguments_are_rejected(hip_lib):
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_2D
    m, s, f = make_mesh_2D(1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    with pytest.raises(A.KnpError):
        dev.upload(A.F_PHI_M, np.zeros(dev.nf + 1))                 # out of bounds
    with pytest.raises(A.KnpError):
        dev.emi_apply(A.F_X, A.F_X)                                 # in-place apply
    with pytest.raises(A.KnpError):
        dev.emi_apply(A.F_PHI_M, A.F_Y)                             # facet field is not a nodal vector
    with pytest.raises(A.KnpError):
        dev.set_params(0.02, 0.0, 96485, 8.314, 300, 200, 40, 40, [1, -1, 1], np.ones((3, dev.nc)))     # dt = 0
    with pytest.raises(A.KnpError):
        dev.set_params(0.02, 1e-4, 96485, 8.314, 300, 200, 40, 40, [1, 0, 1], np.ones((3, dev.nc)))     # z = 0
    assert dev.apply_variant(0) == 0 and dev.apply_variant(1) == 0      # 2D: coordinate-path kernels
    assert dev.apply_variant(2) < 0                                     # no such operator
    dev.close()
    bad = m.facet_cells.copy()
    bad[0, 0] = m.num_cells() + 7

    class Fake:
        pass
    fm = Fake()
    fm.coords, fm.cells, fm.gdim, fm.facet_cells, fm.facet_local = m.coords, m.cells, 2, bad, m.facet_local
    fm.cell_midpoints = m.cell_midpoints
    fm.cell_facets = m.cell_facets
    with pytest.raises((A.KnpError, IndexError)):
        A.Device(fm, s.array(), f.array(), [1], 3, reorder=False)


def test_results_are_bitwise_reproducible(hip_lib):
    """Atomics-free applies and fixed-order reductions: two identical solves give identical bits
    (race detection by deterministic double run)."""
    from knpemidg import _abi as A
    m, s, f = small_3d()
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    synthetic_state(pb)
    outs = []
    for _ in range(2):
        dev = device_for(pb)
        push_state(dev, pb)
        dev.update_kappa(); dev.emi_rhs()
        dev.upload(A.F_PHI, np.zeros(pb.ndof))
        n1, _ = dev.emi_solve(1e-9, maxit=20000)
        dev.update_dnphi(); dev.knp_rhs()
        n2, _ = dev.knp_solve(1e-10, maxit=2000)
        outs.append((dev.download(A.F_PHI), dev.download(A.F_C), n1, tuple(n2)))
        dev.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2:] == outs[1][2:]


def test_membrane_tag_without_facets(hip_lib):
    """A membrane model whose tag marks no facet on this rank (e.g. an x-slab in front of the axons): zero ODE nodes,
    every PDE<->ODE exchange and the ODE step must be no-ops and the time loop must still run."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
    from idealized_common import SolverIdealized, physical_setup, solver_parameters, Constant
    from knpemidg.models import mm_hh, mm_hh_no_stim
    mesh, sub, surf = small_3d()
    params, ion_list, stim = physical_setup()
    S = SolverIdealized(params, ion_list)
    S.verbose = False
    S.setup_domain(mesh, sub, surf)
    S.setup_parameters(); S.setup_FEM_spaces()
    S.setup_membrane_model(stim, {1: mm_hh, 2: mm_hh_no_stim})          # tag 2 does not occur in this mesh
    assert S.mem_models[1]['ode'].nodes == 0 and S.mem_models[1]['ode'].on_device
    assert S.mem_models[1]['ode'].states.shape == (0, 4)
    S._unpack_solver_params(solver_parameters(3, 0))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = Constant(0.0)
    for k in range(2):
        S.step_membrane_models(k)
        S.solve_for_time_step(k, t)
    assert np.isfinite(S.phi.array()).all() and np.isfinite(S.c.array()).all()
    S.dev.close()


def test_p2_edge_cases_and_argument_checks(hip_lib):
    """DG-P2 path on the smallest mesh (two tets, one interior facet, single solved species), plus the argument checks of
    knp_set_tabulation / knp_amg_columns and the loud failure when a tabulation slot is missing."""
    import ctypes as C
    from knpemidg import _abi as A
    from knpemidg.mesh import Mesh, MeshFunction
    coords = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1.0]]) * 1e-6
    m = Mesh(coords, np.array([[0, 1, 2, 3], [1, 2, 3, 4]]))
    s = MeshFunction(m, 3, np.array([0, 1]))
    f = MeshFunction(m, 2, 0)
    f.array()[m.facet_cells[:, 1] >= 0] = 1                        # the interior facet is a membrane
    pb = ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    x = synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.update_dnphi()
    Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
    dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < 1e-11
    dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
    y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
    for k in range(pb.N_ions):
        assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < 1e-11
    dev.emi_rhs(); dev.knp_rhs()
    assert relerr(dev.download(A.F_B_EMI), b) < 1e-11
    # argument checks
    w = np.ones(1); B = np.ones(10); dB = np.zeros(40)
    p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert dev.lib.knp_set_tabulation(dev.ctx, 99, 1, 1, p(w), p(B), p(dB)) != 0          # unknown slot
    assert dev.lib.knp_set_tabulation(dev.ctx, 0, 4, 1, p(w), p(B), p(dB)) != 0           # cell rule with facet layout
    assert dev.lib.knp_set_tabulation(dev.ctx, 3, 1, 1, p(w), p(B), p(dB)) != 0           # facet rule with cell layout
    assert dev.lib.knp_amg_columns(dev.ctx, 0, 2) != 0                                    # EMI hierarchy is single-column
    assert dev.lib.knp_amg_columns(dev.ctx, 1, 99) != 0
    dev.close()
    # a degree-2 context whose tabulations were never uploaded must refuse to integrate the right-hand sides (the operator
    # applies are matrix-free and need none)
    ctx = C.c_void_p()
    from knpemidg._abi import load, _p, _f64p, _i32p, _u32p, _i8p
    lib = load()
    cells = np.ascontiguousarray(m.cells, dtype=np.int32)
    mt = np.array([1], dtype=np.uint32)
    rc = lib.knp_ctx_create(C.byref(ctx), 0, 3, 2, 3, m.num_vertices(), 2, 2, m.num_facets(), _p(np.ascontiguousarray(m.coords), _f64p),
                            _p(cells, _i32p), _p(s.array().astype(np.uint32), _u32p),
                            _p(np.ascontiguousarray(m.facet_cells, dtype=np.int32), _i32p),
                            _p(np.ascontiguousarray(m.facet_local, dtype=np.int8), _i8p), _p(f.array().astype(np.uint32), _u32p), 1,
                            _p(mt, _u32p))
    assert rc == 0
    assert lib.knp_update_kappa(ctx) == 0
    assert lib.knp_emi_rhs(ctx) != 0 and b"tabulation" in lib.knp_last_error(ctx)
    lib.knp_ctx_destroy(ctx)


def test_device_ode_matches_lsoda_oracle(hip_lib):
    """k_ode_step (batched Dormand-Prince 5(4), csrc/ode.hip) against the ORACLE's membrane step: one scipy-LSODA call per
    facet at the reference's rtol 1e-8 / atol 0 on the restated mm_hh.py right-hand side (oracle/membrane_oracle.py,
    reference membrane.py:98-114) -- rows with spatially varying K_e, Na_i, Nernst potentials and a stimulus on part of
    them, 25 steps through the upstroke.  Tolerance: both integrate to rtol 1e-8 per step -> 1e-6 after 25 steps."""
    import membrane_oracle as mo
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace, FacetFunction
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh, mm_hh_no_stim
    m, s, f = make_mesh_2D(1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    Q = FacetSpace(m)
    rng = np.random.default_rng(11)
    fields = {'K_e': 3.32 * (1 + 0.1 * rng.uniform(-1, 1, Q.dim())), 'Na_i': 12.8 * (1 + 0.1 * rng.uniform(-1, 1, Q.dim())),
              'E_K': -0.0936 + 2e-3 * rng.uniform(-1, 1, Q.dim()), 'E_Na': 0.0533 + 2e-3 * rng.uniform(-1, 1, Q.dim())}
    locator = lambda x: x[0] < 20e-6
    for ode, stim in ((mm_hh, True), (mm_hh_no_stim, False)):
        mm = MembraneModel(ode, facet_f=f, tag=1, V=Q)
        mm.set_parameter_values({'Cm': lambda x: 0.02})
        assert mm.attach_device(dev)
        for name, val in fields.items():
            mm.set_parameter(name, FacetFunction(Q, val))
        n = mm.nodes
        st = np.tile(mo.hh_init_states(), (n, 1))
        pr = np.tile(mo.hh_init_parameters(), (n, 1))
        pr[:, mo.P_IDX['Cm']] = 0.02
        for name, val in fields.items():
            pr[:, mo.P_IDX[name]] = val[mm.indices]
        mask = np.fromiter(map(locator, mm.dof_locations), dtype=bool, count=n)
        assert 0 < mask.sum() < n
        for k in range(25):
            mm.step_lsoda(dt=1e-4, stimulus={'stim_amplitude': 40.0}, stimulus_locator=locator)
            mo.step_lsoda(st, pr, k * 1e-4, 1e-4, stim=stim, stimulus={'stim_amplitude': 40.0}, stimulus_mask=mask)
        sd, pd = mm.states, mm.parameters
        assert np.abs(sd - st).max() < 1e-6 * np.abs(st).max(), (stim, np.abs(sd - st).max())
        cur = slice(mo.P_IDX['I_ch_Na'], mo.P_IDX['I_ch_K'] + 1)
        assert np.abs(pd[:, cur] - pr[:, cur]).max() < 1e-5 * np.abs(pr[:, cur]).max()
        if stim:
            assert st[mask, 3].max() > -0.05 and st[~mask, 3].max() < -0.06     # stimulated rows depolarise, the others rest
    dev.close()


@pytest.mark.parametrize("which", ["hh_emix", "glial", "leak"])
def test_device_ode_other_models_match_lsoda_oracle(hip_lib, which):
    """k_ode_step models 3 / 4 / 5 -- the EMIx neuron and glial membranes of BASELINE configs[4] (reference:
    examples/emix-simulations/mm_hh.py:118-161, mm_glial.py:117-170; cm / ms / mV) and the passive leak membrane of the
    rat-neuron example (examples/rat-neuron/mm_leak.py:107-133; SI) -- against the ORACLE's independent restatement of the
    same reference text stepped by scipy LSODA per facet at rtol 1e-8 / atol 0 (oracle/membrane_oracle.py; reference
    membrane.py:108-112): rows with varying K_e, Na_i, E_K, E_Na, a stimulus on part of them, 25 steps.
    Tolerance as for the HH test: both integrate to rtol 1e-8 per step -> 1e-6 (states), 1e-5 (currents) after 25 steps."""
    import membrane_oracle as mo
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace, FacetFunction
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh_emix, mm_glial, mm_leak
    ode = {"hh_emix": mm_hh_emix, "glial": mm_glial, "leak": mm_leak}[which]
    si = which == "leak"
    m, s, f = make_mesh_2D(1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    Q = FacetSpace(m)
    rng = np.random.default_rng(23)
    u = lambda: rng.uniform(-1, 1, Q.dim())
    mV = 1e-3 if si else 1.0
    fields = {'K_e': 3.32 * (1 + 0.2 * u()), 'Na_i': 12.8 * (1 + 0.1 * u()),
              'E_K': (-93.6 + 3.0 * u()) * mV, 'E_Na': (53.3 + 2.0 * u()) * mV}
    Cm, dt, amp = (0.02, 1e-4, 40.0) if si else (2.0, 0.1, 5.0)               # run_rat_neuron / run_EMIx_simulation settings
    locator = lambda x: x[0] < 20e-6
    init_s, init_p, _, pidx, iV = mo.MODELS[which]
    assert ode.parameter_indices('I_ch_Na') == pidx['I_ch_Na'] and ode.state_indices('V') == iV
    assert np.array_equal(ode.init_state_values(), init_s()) and np.array_equal(ode.init_parameter_values(), init_p())
    mm = MembraneModel(ode, facet_f=f, tag=1, V=Q)
    mm.set_parameter_values({'Cm': lambda x: Cm})
    assert mm.attach_device(dev) and mm.on_device
    for name, val in fields.items():
        mm.set_parameter(name, FacetFunction(Q, val))
    n = mm.nodes
    st = np.tile(init_s(), (n, 1))
    pr = np.tile(init_p(), (n, 1))
    pr[:, pidx['Cm']] = Cm
    for name, val in fields.items():
        pr[:, pidx[name]] = val[mm.indices]
    mask = np.fromiter(map(locator, mm.dof_locations), dtype=bool, count=n)
    assert 0 < mask.sum() < n
    v0 = st[:, iV].copy()
    for k in range(25):
        mm.step_lsoda(dt=dt, stimulus={'stim_amplitude': amp}, stimulus_locator=locator)
        mo.step_lsoda_model(which, st, pr, k * dt, dt, stimulus={'stim_amplitude': amp}, stimulus_mask=mask)
    sd, pd = mm.states, mm.parameters
    assert np.abs(sd - st).max() < 1e-6 * np.abs(st).max(), np.abs(sd - st).max()
    cur = [pidx['I_ch_Na'], pidx['I_ch_K']]
    assert np.abs(pd[:, cur] - pr[:, cur]).max() < 1e-5 * np.abs(pr[:, cur]).max()
    assert np.abs(st[:, iV] - v0).max() > 1e-3 * np.abs(v0).max()             # the rows moved: the comparison is not of a rest state
    if which != "glial":                                                     # the glial model has no stimulus term
        assert np.abs(st[mask, iV] - v0[mask]).max() > 2 * np.abs(st[~mask, iV] - v0[~mask]).max()
    dev.close()


def test_calibration_system_matches_lsoda_oracle_and_reaches_the_emix_initial_state(hip_lib):
    """The ODE-only calibration run of BASELINE configs[4] ("calibrated ICs"; reference:
    examples/emix-simulations/run_calibration.py:13-90 with mm_calibration.py:143-255) on the device (k_ode_step model 6):
    (1) 25 steps from perturbed states against the oracle's scalar restatement stepped by scipy LSODA (rtol 1e-8 / atol 0) ->
    1e-6; (2) the driver run to stationarity reproduces the initial values the reference hard-codes in mm_hh.py:11-14 (m, h, n,
    phi_M of the neuron: all printed digits) and, to the 3-4 digits that survive the reference's own parameter revisions, the
    concentrations of run_EMIx_simulation.py:76-84."""
    import membrane_oracle as mo
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "emix_simulations"))
    import run_calibration as rc
    from knpemidg import _abi as A
    from knpemidg.mesh import RectangleMesh, MeshFunction
    from knpemidg.functions import FacetSpace
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_calibration as ode
    assert np.array_equal(ode.init_state_values(), mo.calibration_init_states())
    assert np.array_equal(ode.init_parameter_values(), mo.calibration_init_parameters())
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), 2, 2)
    facet_f = MeshFunction(mesh, 1, 0)
    dev = A.Device(mesh, np.zeros(mesh.num_cells(), dtype=np.uint32), facet_f.array(), (), 3)
    mm = MembraneModel(ode, facet_f=facet_f, tag=0, V=FacetSpace(mesh))
    assert mm.attach_device(dev)
    n = mm.nodes
    rng = np.random.default_rng(4)
    st = np.tile(mo.calibration_init_states(), (n, 1)) * (1 + 0.02 * rng.uniform(-1, 1, (n, 11)))
    pr = np.tile(mo.calibration_init_parameters(), (n, 1))
    pr[:, mo.CALIBRATION_P_IDX["stim_amplitude"]] = 3.0
    mm.states = st.copy()
    for k in range(25):
        mm.step_lsoda(dt=0.1, stimulus={'stim_amplitude': 3.0})
        mo.step_lsoda_plain(mo.calibration_rhs, st, pr, k * 0.1, 0.1)
    sd = mm.states
    assert np.abs(sd - st).max() < 1e-6 * np.abs(st).max(), np.abs(sd - st).max()
    assert np.abs(st[:, 3] + 74.38).max() > 1.0                        # the stimulated neuron moved: not a comparison of rest states
    dev.close()
    # (2) the driver; 40 000 steps of 0.1 ms = 4 s (stationary to 1e-9 after ~3 s; the reference runs 100 000)
    out, states = rc.calibrate(40000, verbose=False)
    assert np.abs(states - states[0]).max() == 0.0                     # every node integrates the same system
    hh = dict(m_init=0.016651023270342777, h_init=0.8541791472445746, n_init=0.18821645700362638, phi_M_n_init=-74.3848784437955)
    for key, ref in hh.items():                                        # examples/emix-simulations/mm_hh.py:11-14
        assert abs(out[key] - ref) < 2e-7 * abs(ref), (key, out[key], ref)
    near = dict(phi_M_g_init=-83.08511451850003, K_e_init=3.3236967382613933, K_n_init=124.15397583492471, K_g_init=102.75563828644862,
                Na_e_init=100.71925900028181, Na_n_init=12.838513108606818, Na_g_init=12.39731187972181)
    for key, ref in near.items():                                      # mm_glial.py:11, run_EMIx_simulation.py:76-84
        assert abs(out[key] - ref) < 2e-4 * abs(ref), (key, out[key], ref)


def test_stimulus_is_reimposed_every_step(hip_lib):
    """The reference overwrites the stimulus parameters on the masked rows at the start of every step_lsoda call
    (membrane.py:98-104): a hook that rewrites the whole parameter table between steps must not lose the stimulus."""
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh
    m, s, f = make_mesh_2D(0)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    Q = FacetSpace(m)
    res = []
    for clobber in (False, True):
        mm = MembraneModel(mm_hh, facet_f=f, tag=1, V=Q)
        mm.set_parameter_values({'Cm': lambda x: 0.02, 'K_e': lambda x: 3.32, 'Na_i': lambda x: 12.8,
                                 'E_K': lambda x: -0.0936, 'E_Na': lambda x: 0.0533})
        assert mm.attach_device(dev)
        for k in range(8):
            if clobber:
                p = mm.parameters
                p[:, mm_hh.parameter_indices('stim_amplitude')] = 0.0        # a hook wipes the stimulus column
                mm.parameters = p
            mm.step_lsoda(dt=1e-4, stimulus={'stim_amplitude': 40.0}, stimulus_locator=lambda x: x[0] < 20e-6)
        res.append(mm.states)
    assert np.array_equal(res[0], res[1])
    assert res[0][:, 3].max() > -0.06
    dev.close()


def test_set_params_between_solves_refreshes_lagged_preconditioner_data(hip_lib):
    """dt (and with it C_phi and the KNP mass term) changes through set_params between solves: the lagged block-Jacobi
    inverses AND the Chebyshev bound lambda_max(Binv A) built on them must be rebuilt, otherwise the two-step Chebyshev
    smoother amplifies the top modes and BiCGStab stagnates.  Both steps must converge to the oracle's solution."""
    from knpemidg import _abi as A
    m, s, f = small_3d((10, 4, 4))
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    z = [ion["z"] for ion in pb.ions]
    D = np.stack([ion["D"] for ion in pb.ions])
    its, its_fresh = [], []
    for dt in (1e-4, 1e-7, 1e-2):
        pb.dt = dt
        pb.C_phi = pb.C_M / dt
        ko.solve_emi(pb, direct=True)
        c0 = pb.c.copy()
        ref = ko.solve_knp(pb, direct=True)
        pb.c = c0
        fresh = device_for(pb)                           # a context that has never seen another dt
        for d, out in ((dev, its), (fresh, its_fresh)):
            d.set_params(pb.C_M, dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=True)
            push_state(d, pb)
            d.update_dnphi(); d.knp_rhs()
            niter, res = d.knp_solve(1e-12, maxit=5000)
            assert relerr(d.download(A.F_C).reshape(pb.c.shape), ref) < 1e-8, (dt, niter, res)
            out.append(max(niter))
        fresh.close()
    # the long-lived context converges like a fresh one at every dt (stale inverses / bounds would cost many more iterations)
    assert all(a <= 1.3 * b + 2 for a, b in zip(its, its_fresh)), (its, its_fresh)
    dev.close()


def test_stale_projection_result_raises(hip_lib):
    """pcws_constant_project results live in a few rotating device scratch slots: a result that has been overwritten by
    later projections raises instead of silently aliasing them; results consumed in time are distinct."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))
    from idealized_common import make_solver
    from knpemidg.utils import pcws_constant_project, plus, minus
    S = make_solver(dim=2, resolution=1)
    a = pcws_constant_project(plus(S.c_prev_k.split()[0], S.n_g), S.Q)
    b = pcws_constant_project(minus(S.ion_list[-1]['c'], S.n_g), S.Q)
    va, vb = a.array(), b.array()
    mem = np.nonzero(S.surfaces.array() == 1)[0]
    assert np.allclose(va[mem], 3.3236967382705265) and np.allclose(vb[mem], 12.838513108648856)   # K_e, Na_i: both alive
    for _ in range(4):
        pcws_constant_project(plus(S.c_prev_k.split()[1], S.n_g), S.Q)
    with pytest.raises(RuntimeError, match="overwritten"):
        a.array()
    S.dev.close()


def test_fp64_mfma_probe_matches_fma_chain(hip_lib):
    """The DG-P2 facet-quadrature contraction as v_mfma_f64_16x16x4_f64 tiles (diagnostic variant, csrc/apply_p2.hip) gives
    the same numbers as the per-thread FMA chain of the product kernels -- and as the numpy formula -- on random inputs,
    including a ragged column count (partial last tile)."""
    import p2_formulation as pf
    from knpemidg.mesh import make_mesh_2D
    m, s, f = make_mesh_2D(0)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    rng = np.random.default_rng(7)
    ncol = 16 * 37 + 5
    a = rng.uniform(-1, 1, size=(ncol, 26))
    a[:, 6:18] = rng.uniform(1, 2, size=(ncol, 12))             # kappa > 0
    a[:, 24:] = rng.uniform(0.5, 1.5, size=(ncol, 2))
    out0, _ = dev.probe_facet_contraction(0, a, reps=2)
    out1, _ = dev.probe_facet_contraction(1, a, reps=2)
    T = pf.load_tables(3)
    ref = np.zeros((ncol, 9))
    for q in range(len(T["WE"])):
        psi, lam, w = T["PSIE"][q], T["LAME"][q], T["WE"][q] * a[:, 24]
        ko_, kn_, jq = a[:, 6:12] @ psi, a[:, 12:18] @ psi, a[:, 0:6] @ psi
        flux = w * (a[:, 25] * (ko_ + kn_) * jq - 0.5 * (ko_ * (a[:, 18:21] @ lam) + kn_ * (a[:, 21:24] @ lam)))
        t = -0.5 * w * ko_ * jq
        ref[:, :6] += flux[:, None] * psi[None, :]
        ref[:, 6:] += t[:, None] * lam[None, :]
    assert relerr(out0, ref) < 1e-13
    assert relerr(out1, ref) < 1e-13
    dev.close()
