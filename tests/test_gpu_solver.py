"""GPU solves and the full splitting step vs the oracle (direct sparse solves on small meshes).
Tolerances: converged c <= 1e-6 relative, mean-free phi <= 1e-4 relative at the
reference's rtol (1e-5 / 1e-7); tighter here because both sides are solved tightly."""
import os
import sys

import numpy as np
import pytest

import knpemi_oracle as ko
from common import synthetic_state, device_for, push_state, relerr, small_3d, mean_free

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries"))


def _small_problems():
    from knpemidg.mesh import make_mesh_2D
    out = {}
    m, s, f = make_mesh_2D(0)
    out["2D"] = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    m, s, f = small_3d()
    out["3D"] = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    m, s, f = make_mesh_2D(0)
    out["2D_P2"] = ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    m, s, f = small_3d((6, 3, 3))
    out["3D_P2"] = ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    return out


@pytest.fixture(scope="module", params=["2D", "3D", "2D_P2", "3D_P2"])
def case(request, hip_lib):
    from knpemidg import _abi as A
    pb = _small_problems()[request.param]
    synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    yield pb, dev, A
    dev.close()


def test_emi_solve(case):
    pb, dev, A = case
    dev.update_kappa()
    dev.emi_rhs()
    dev.upload(A.F_PHI, np.zeros(pb.ndof))
    niter, res = dev.emi_solve(1e-11, maxit=20000)
    phi = dev.download(A.F_PHI)
    ref = ko.solve_emi(pb, direct=True).copy()
    a, b = mean_free(phi, pb.geom.vol), mean_free(ref, pb.geom.vol)
    assert relerr(a, b) < 1e-7, (niter, res)
    assert niter > 0


def test_knp_solve(case):
    pb, dev, A = case
    # phi from the oracle's EMI solve so that both sides use the same potential
    ko.solve_emi(pb, direct=True)
    dev.upload(A.F_PHI, pb.phi)
    dev.update_dnphi()
    dev.knp_rhs()
    niter, res = dev.knp_solve(1e-13, maxit=5000)
    c = dev.download(A.F_C).reshape(pb.c.shape)
    c0 = pb.c.copy()
    ref = ko.solve_knp(pb, direct=True)
    pb.c = c0
    assert relerr(c, ref) < 1e-9, (niter, res)
    assert all(n >= 5 for n in niter)          # ksp_min_it 5 (solver.py:686)


def test_knp_early_stop_below_the_iteration_floor(case):
    """knp_knp_early_stop: from a converged state a BiCGStab solve runs its floor of min_it iterations with factor 0 (the reference's
    ksp_min_it, solver.py:686) and stops after ONE iteration when its residual is already the factor under the tolerance; a solve
    that is not that far converges as before; factors outside [0, 1) are refused."""
    pb, dev, A = case
    ko.solve_emi(pb, direct=True)
    dev.upload(A.F_PHI, pb.phi)
    dev.update_dnphi()
    dev.knp_rhs()
    dev.knp_solve(1e-13, maxit=5000)
    c_ref = dev.download(A.F_C)
    near = c_ref * (1.0 + 1e-10 * np.random.default_rng(7).uniform(-1.0, 1.0, size=c_ref.shape))    # a quiet step's initial guess
    dev.upload(A.F_C, near)
    dev.knp_early_stop(0.0)
    niter, res = dev.knp_solve(1e-6, maxit=100, min_it=5)
    assert all(n == 5 for n in niter), (niter, res)
    dev.upload(A.F_C, near)
    dev.knp_early_stop(0.01)
    niter, res = dev.knp_solve(1e-6, maxit=100, min_it=5)
    assert all(n == 1 for n in niter), (niter, res)
    assert np.all(res[:, 1] <= 0.01 * 20.0 * 1e-6 * res[:, 2])         # 20: the device's factor on the order-8 density test (csrc/abi.hip)
    assert relerr(dev.download(A.F_C), c_ref) < 1e-9
    # not yet 100x under the tolerance after the first iteration: the solve goes on as before
    dev.upload(A.F_C, c_ref * (1.0 + 1e-3 * np.random.default_rng(8).uniform(-1.0, 1.0, size=c_ref.shape)))
    niter, res = dev.knp_solve(1e-6, maxit=100, min_it=5)
    assert all(n >= 2 for n in niter) and np.all(res[:, 1] <= 20.0 * 1e-6 * res[:, 2]), (niter, res)
    with pytest.raises(A.KnpError):
        dev.knp_early_stop(1.5)
    dev.knp_early_stop(0.0)
    dev.upload(A.F_C, pb.c)                               # the fixture's state for the tests behind this one


@pytest.mark.parametrize("restart", [30, 8])
def test_knp_solve_gmres(case, restart):
    """The reference's KNP Krylov method, restarted GMRES (ksp_type gmres, restart 30: solver.py:684-701), on the device: same
    preconditioner and stopping test as the default BiCGStab, same converged concentrations as the oracle's direct solve; restart 8
    forces several restart cycles (solution update, true residual, new cycle)."""
    pb, dev, A = case
    ko.solve_emi(pb, direct=True)
    dev.upload(A.F_PHI, pb.phi)
    dev.upload(A.F_C, pb.c)
    dev.update_dnphi()
    dev.knp_rhs()
    dev.set_knp_krylov("gmres", restart)
    try:
        niter, res = dev.knp_solve(1e-13, maxit=5000)
    finally:
        dev.set_knp_krylov("bicgstab")
    c = dev.download(A.F_C).reshape(pb.c.shape)
    c0 = pb.c.copy()
    ref = ko.solve_knp(pb, direct=True)
    pb.c = c0
    assert relerr(c, ref) < 1e-9, (niter, res)
    assert all(n >= 5 for n in niter)
    with pytest.raises(Exception):
        dev.set_knp_krylov("gmres", 31)


@pytest.mark.parametrize("names,shared", [(("K", "Cl", "X", "Na"), True), (("K", "Cl", "X", "Na"), False), (("K", "Cl"), True)])
def test_knp_solve_with_other_species_counts(hip_lib, monkeypatch, names, shared):
    """KNP solves with ONE and with THREE solved species through the auxiliary-space preconditioner: three species sharing one
    hierarchy ride the V-cycle as an odd number of right-hand-side columns (not interleaved in pairs), or get one hierarchy each;
    the converged concentrations are the oracle's direct solve."""
    from knpemidg import _abi as A, amg
    from knpemidg.mesh import make_mesh_3D
    monkeypatch.setenv("KNP_AMG_SHARED", "1" if shared else "0")
    m, s, f = make_mesh_3D(0, n_axons=1)
    P = ko.idealized_params()
    nc = m.num_cells()
    z = dict(P["z"], X=1.0)        # (a divalent ion in the seeded random potential of +-70 mV per node makes the operator so
    Dc = dict(P["D"], X=1.6e-9)    #  advection-dominated that BiCGStab breaks down with or without the hierarchy: not what is tested)
    ions = [dict(name=n, z=z[n], D=np.full(nc, Dc[n])) for n in names]
    pb = ko.Problem(m, s.array().astype(np.int64), f.array(), 1, ions, P, membrane_tags=(1,))
    rng = np.random.default_rng(7)
    pb.c = rng.uniform(80.0, 120.0, size=pb.c.shape)
    pb.c_prev_n = pb.c * (1 + 1e-3 * rng.uniform(-1, 1, size=pb.c.shape))
    pb.c_elim = rng.uniform(80.0, 120.0, size=pb.c_elim.shape)
    synthetic_state(pb)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        cs = amg.ConformingSpace(m, f.array(), (1,))
        groups = amg.build_knp_groups(cs, None, s.array(), [{0: Dc[n], 1: Dc[n]} for n in names[:-1]], pb.dt, 1)
        assert len(groups) == (1 if shared else len(names) - 1)
        for members, levels in groups:
            dev.amg_upload(1 + members[0], cs.dof, levels, ncol=len(members))
            for k in members[1:]:
                dev.amg_clear(1 + k)
        dev.update_dnphi()
        dev.knp_rhs()
        niter, res = dev.knp_solve(1e-13, maxit=2000)
        c = dev.download(A.F_C).reshape(pb.c.shape)
        ref = ko.solve_knp(pb, direct=True)
        assert relerr(c, ref) < 1e-9, (niter, res)
        assert max(niter) < 80                      # the hierarchy is active (block-Jacobi alone: hundreds)
    finally:
        dev.close()


@pytest.mark.parametrize("dim,degree", [(2, 1), (3, 1), (2, 2), (3, 2)])
def test_active_time_loop(hip_lib, dim, degree):
    """Three splitting steps with HH membranes + stimulus: GPU Solver vs oracle stepping fed with the
    same ODE outputs (the ODE step is adjacent to the hot path).  degree 2 is the
    `Solver(degree_emi=2, degree_knp=2)` path of BASELINE configs[2]."""
    from common_examples import make_solver, solver_parameters, Constant
    from knpemidg.mesh import make_mesh_2D
    mt = make_mesh_2D(0) if dim == 2 else small_3d((8, 4, 4) if degree == 1 else (6, 3, 3))
    S = make_solver(dim=dim, resolution=0, n_axons=1, mesh_tuple=mt, degree=degree)
    sp = solver_parameters(dim, 0, max_it_emi=50000)
    sp = sp._replace(rtol_emi=1e-10, rtol_knp=1e-12)
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    pb = ko.build_idealized(mt[0], mt[1].array(), mt[2].array(), p=degree, membrane_tags=(1,))
    pb.phi_M[:] = 0.0
    t = Constant(0.0)
    for k in range(3):
        S.step_membrane_models(k)
        # feed the oracle with the ODE outputs
        pb.phi_M = S.phi_M_prev_PDE.array().copy()
        for ion in pb.ions:
            pb.I_ch[ion["name"]] = S.mem_models[0]['I_ch_k'][ion["name"]].array().copy()
        S.solve_for_time_step(k, t)
        E = ko.solve_for_time_step(pb, direct=True)
        vol = pb.geom.vol
        assert relerr(mean_free(S.phi.array(), vol), mean_free(pb.phi, vol)) < 1e-6
        assert relerr(S.c.array(), pb.c) < 1e-8
        assert relerr(S.ion_list[-1]['c'].array(), pb.c_elim) < 1e-8
        assert relerr(S.phi_M_prev_PDE.array()[pb.mem], pb.phi_M[pb.mem]) < 1e-6
        for ki, ion in enumerate(S.ion_list):
            # E_k = RT/(F z) avg ln(c_e/c_i) amplifies the (1e-9 relative) solver error in c where c_e ~ c_i
            assert relerr(ion['E'].array()[pb.mem], E[ion['name']]) < (1e-7 if degree == 1 else 1e-6)
    # the stimulus must have moved the membrane potential on the stimulated part
    assert np.abs(S.phi_M_prev_PDE.array()[pb.mem] + 0.0743861).max() > 1e-5
    assert abs(float(t) - 3e-4) < 1e-12


@pytest.mark.parametrize("switch", ["KNP_FUSE_RESTRICT", "KNP_FUSE_FIRST0", "KNP_FUSE_CG_RESTRICT"])
def test_fused_chebyshev_restriction_equals_two_passes(hip_lib, monkeypatch, switch):
    """The second Chebyshev block-Jacobi step fused with stage 1 of the tile-wise restriction (k_bj_cheb2_restrict, default) against
    the two separate kernels (KNP_FUSE_RESTRICT=0), and the finest conforming level's first Chebyshev update written by stage 2 of the
    restriction -- into the buffers the captured V-cycle reads -- against its own launch (KNP_FUSE_FIRST0=0), and the PCG update fused with
    stage 1 of the restriction of the new residual (k_cg_update_restrict) against two kernels (KNP_FUSE_CG_RESTRICT=0): same preconditioner, so the
    same iteration counts and, to rounding, the same fields after three stimulated steps of the 4-axon mesh with its AMG hierarchies
    (PCG for EMI, BiCGStab and GMRES for KNP).  The coarse-size limit leaves a level below the finest one on this small mesh."""
    monkeypatch.setenv("KNP_AMG_MAXCOARSE", "300")
    from idealized_common import make_solver, solver_parameters, Constant
    out = {}
    for krylov in ("bicgstab", "gmres"):
        for fused in ("1", "0"):
            monkeypatch.setenv(switch, fused)
            S = make_solver(dim=3, resolution=0, n_axons=4)
            sp = solver_parameters(3, 0)
            if switch == "KNP_FUSE_CG_RESTRICT":
                # the PCG update fused with the restriction of the new residual (k_cg_update_restrict) exists for EMI preconditioners
                # without the DG-level Chebyshev step, which a mesh of this size would keep: switched off explicitly
                from collections import namedtuple
                sp = namedtuple("solver_params", sp._fields + ("emi_dg_chebyshev",))(*sp, False)
            S._unpack_solver_params(sp)
            S.save_fields = S.save_solver_stats = False
            S.splitting_scheme = True
            S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
            S.dev.set_knp_krylov(krylov)
            assert S.use_amg
            t = Constant(0.0)
            for k in range(3):
                S.step_membrane_models(k)
                S.solve_for_time_step(k, t)
            out[(krylov, fused)] = (S.phi.array().copy(), S.c.array().copy(), list(S.emi_niter), [list(n) for n in S.knp_niter])
            S.dev.close()
        a, b = out[(krylov, "1")], out[(krylov, "0")]
        assert a[2] == b[2] and a[3] == b[3], (a[2:], b[2:])
        assert relerr(a[0], b[0]) < 1e-10 and relerr(a[1], b[1]) < 1e-12


def test_amg_vcycle_matches_reference_and_cuts_iterations(hip_lib):
    """Auxiliary-space AMG: (1) Ac assembled by knpemidg.amg equals P^T A P of the oracle matrix; (2) PCG with the
    device V-cycle converges to the same phi in far fewer iterations than block-Jacobi alone."""
    import scipy.sparse as sp
    from knpemidg import _abi as A, amg
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(0, n_axons=1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    synthetic_state(pb)
    Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
    cs = amg.ConformingSpace(m, f.array(), (1,))
    Ac = cs.stiffness(pb.kappa(), membrane=(pb.mem, pb.C_phi))
    P = sp.coo_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), cs.dof.ravel())), shape=(pb.ndof, cs.n)).tocsr()
    ref = (P.T @ Aemi @ P).tocsr()
    assert abs(Ac - ref).max() < 1e-12 * abs(ref).max()
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.emi_rhs()
    dev.upload(A.F_PHI, np.zeros(pb.ndof))
    n_bj, _ = dev.emi_solve(1e-8, maxit=50000)
    phi_bj = dev.download(A.F_PHI)
    levels = amg.build_hierarchy(Ac)
    dev.amg_upload(0, cs.dof, levels)
    dev.upload(A.F_PHI, np.zeros(pb.ndof))
    n_amg, res = dev.emi_solve(1e-8, maxit=2000)
    phi_amg = dev.download(A.F_PHI)
    vol = pb.geom.vol
    assert relerr(mean_free(phi_amg, vol), mean_free(phi_bj, vol)) < 1e-5
    assert n_amg * 10 < n_bj, (n_amg, n_bj)
    dev.close()


def test_device_ode_matches_host(hip_lib):
    """Batched HIP Dormand-Prince integrator (csrc/ode.hip) vs the host numpy integrator on the HH model with a
    spatially varying stimulus: same pair, controller and tolerances -> agreement far below rtol 1e-8 * steps."""
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace, FacetFunction
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh
    m, s, f = make_mesh_2D(1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    Q = FacetSpace(m)
    models = []
    for on_dev in (False, True):
        mm = MembraneModel(mm_hh, facet_f=f, tag=1, V=Q)
        mm.set_parameter_values({'Cm': lambda x: 0.02})
        if on_dev:
            assert mm.attach_device(dev)
        for name, val in (('K_e', 3.32), ('Na_i', 12.8), ('E_K', -0.0936), ('E_Na', 0.0533)):
            mm.set_parameter(name, FacetFunction(Q, np.full(Q.dim(), val)))
        models.append(mm)
    for k in range(10):
        for mm in models:
            mm.step_lsoda(dt=1e-4, stimulus={'stim_amplitude': 40.0}, stimulus_locator=lambda x: x[0] < 20e-6)
    sh, sd = models[0].states, models[1].states
    ph, pd = models[0].parameters, models[1].parameters
    assert np.abs(sh - sd).max() < 1e-9 * np.abs(sh).max()
    assert np.abs(ph[:, 8:10] - pd[:, 8:10]).max() < 1e-8 * np.abs(ph[:, 8:10]).max()
    assert sh[:, 3].max() > -0.07                                  # stimulated nodes depolarised
    # ODE -> PDE scatter lands on the right facets
    dev.upload(A.F_PHI_M, np.zeros(Q.dim()))
    from knpemidg.functions import DeviceFacetFunction
    models[1].get_membrane_potential(DeviceFacetFunction(Q, dev, A.F_PHI_M))
    out = dev.download(A.F_PHI_M)
    assert np.allclose(out[models[1].indices], sd[:, 3]) and (np.delete(out, models[1].indices) == 0).all()
    dev.close()


def test_distributed_setup_path_world1(hip_lib):
    """make_distributed_solver with a single rank walks the whole distributed code path except the RCCL calls
    (partition -> local mesh -> owned/ghost device context -> replicated global AMG hierarchy built from the
    tag-wise initial state) and must reproduce the plain single-GPU solver."""
    from common_examples import make_solver, solver_parameters, Constant
    from knpemidg.partition import make_distributed_solver
    from knpemidg.mesh import make_mesh_3D

    class FakeDist:
        @staticmethod
        def broadcast_object_list(lst, src=0):
            return None
    out = []
    for distributed in (False, True):
        mt = make_mesh_3D(0, n_axons=1)
        if distributed:
            S = make_distributed_solver(3, 0, rank=0, world=1, local_rank=0, dist=FakeDist, n_axons=1, mesh_tuple=mt)
        else:
            S = make_solver(dim=3, resolution=0, n_axons=1, mesh_tuple=mt)
        S._unpack_solver_params(solver_parameters(3, 0))
        S.save_fields = S.save_solver_stats = False
        S.splitting_scheme = True
        S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
        t = Constant(0.0)
        for k in range(2):
            S.step_membrane_models(k)
            S.solve_for_time_step(k, t)
        phi = S.phi.array()
        out.append((phi - phi.mean(), S.c.array(), S.phi_M_prev_PDE.array(), list(S.emi_niter)))
        S.dev.close()
    assert relerr(out[1][0], out[0][0]) < 1e-4          # both solved to rtol 1e-5 with (slightly) different AMG setups
    assert relerr(out[1][1], out[0][1]) < 1e-6
    assert relerr(out[1][2], out[0][2]) < 1e-4
    assert max(out[1][3]) < 200 and max(out[0][3]) < 200   # AMG active in both


def test_mms_space_convergence_on_device(hip_lib):
    """The reference's verification test (tests/run_MMS_space.py) on the HIP path: manufactured solution, 2^r x 2^r
    meshes, P1, dt = 1e-10, two steps -> L2 order 2 for all concentrations and the (mean-corrected) potential."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mms"))
    import run_MMS_space as R
    errs = [R.run(r) for r in (3, 4, 5)]
    for key in ("a", "b", "c", "phi"):
        rate = np.log(errs[-2][key] / errs[-1][key]) / np.log(2)
        assert rate > 1.9, (key, rate, errs)
    # same numbers as the oracle on the same problem (independent restatements of the MMS data)
    import mms as omms
    pb = omms.build_space_mms(5, p=1)
    for _ in range(2):
        ko.solve_for_time_step(pb, direct=True)
    ref = omms.l2_errors(pb)
    for key in ("a", "b", "c", "phi"):
        assert abs(errs[-1][key] - ref[key]) < 2e-3 * ref[key], (key, errs[-1][key], ref[key])


def test_mms_space_convergence_p2_on_device(hip_lib):
    """The same manufactured solution with `Solver(degree_emi=2, degree_knp=2)` (the reference reaches P2 exactly this
    way, tests/run_MMS_space.py:194-195): L2 order 3 on the assembled-block P2 path, and the same errors as the oracle."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mms"))
    import run_MMS_space as R
    errs = [R.run(r, degree=2) for r in (2, 3, 4)]
    for key in ("a", "b", "c", "phi"):
        rate = np.log(errs[-2][key] / errs[-1][key]) / np.log(2)
        assert rate > 2.8, (key, rate, errs)
    import mms as omms
    pb = omms.build_space_mms(4, p=2)
    for _ in range(2):
        ko.solve_for_time_step(pb, direct=True)
    ref = omms.l2_errors(pb)
    for key in ("a", "b", "c", "phi"):
        assert abs(errs[-1][key] - ref[key]) < 1e-2 * ref[key], (key, errs[-1][key], ref[key])


def test_picard_variant(hip_lib):
    """solve_for_time_step_picard (solver.py:850-927): converges in a few Picard levels and lands close to the plain
    splitting step for a small time step (both are consistent discretisations of the same coupled step)."""
    from common_examples import make_solver, solver_parameters, Constant
    res = []
    for picard in (False, True):
        S = make_solver(dim=3, resolution=0, n_axons=1, mesh_tuple=small_3d())
        S._unpack_solver_params(solver_parameters(3, 0)._replace(rtol_emi=1e-9, rtol_knp=1e-11))
        S.save_fields = S.save_solver_stats = False
        S.splitting_scheme = True
        S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
        t = Constant(0.0)
        for k in range(2):
            S.step_membrane_models(k)
            (S.solve_for_time_step_picard if picard else S.solve_for_time_step)(k, t)
        res.append((S.c.array(), S.phi_M_prev_PDE.array(), getattr(S, "picard_iters", None)))
        assert abs(float(t) - 2e-4) < 1e-12
        S.dev.close()
    assert all(1 <= n <= 25 for n in res[1][2])
    assert relerr(res[1][0], res[0][0]) < 1e-3
    mem = np.nonzero(res[0][1])[0]
    assert relerr(res[1][1][mem], res[0][1][mem]) < 1e-2


def test_mms_time_convergence_on_device(hip_lib):
    """The reference's second verification test (tests/run_MMS_time.py) on the HIP path: halving dt halves the
    concentration errors (first order: backward Euler + splitting with data at the old time level)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mms"))
    import run_MMS_time as R
    errs = []
    for i in (1, 2, 3):
        e, tend = R.run(i, resolution=4)
        assert abs(tend - 2e-2) < 1e-12
        errs.append(e)
    for key in ("a", "b", "c"):
        r1 = np.log(errs[0][key] / errs[1][key]) / np.log(2)
        r2 = np.log(errs[1][key] / errs[2][key]) / np.log(2)
        assert 0.8 < r2 < 1.3 and 0.7 < r1 < 1.4, (key, r1, r2, errs)


def test_baseline_config1_2d_neuron_one_step(hip_lib):
    """BASELINE configs[0]: 2D idealized single neuron + ECS (make_mesh_2D r=2: 3 968 triangles, 35 712 DoFs), HH
    membrane with stimulus, P1, ONE splitting step: HIP path vs the oracle's assembled-CSR direct solves."""
    from common_examples import make_solver, solver_parameters, Constant
    from knpemidg.mesh import make_mesh_2D
    mt = make_mesh_2D(2)
    assert mt[0].num_cells() == 3968
    S = make_solver(dim=2, resolution=2, mesh_tuple=mt)
    S._unpack_solver_params(solver_parameters(2, 2)._replace(rtol_emi=1e-10, rtol_knp=1e-12))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    pb = ko.build_idealized(mt[0], mt[1].array(), mt[2].array(), membrane_tags=(1,))
    pb.phi_M[:] = 0.0
    t = Constant(0.0)
    S.step_membrane_models(0)
    pb.phi_M = S.phi_M_prev_PDE.array().copy()
    for ion in pb.ions:
        pb.I_ch[ion["name"]] = S.mem_models[0]['I_ch_k'][ion["name"]].array().copy()
    S.solve_for_time_step(0, t)
    E = ko.solve_for_time_step(pb, direct=True)
    vol = pb.geom.vol
    assert relerr(mean_free(S.phi.array(), vol), mean_free(pb.phi, vol)) < 1e-6
    assert relerr(S.c.array(), pb.c) < 1e-8
    assert relerr(S.phi_M_prev_PDE.array()[pb.mem], pb.phi_M[pb.mem]) < 1e-6
    for ion in S.ion_list:
        assert relerr(ion['E'].array()[pb.mem], E[ion['name']]) < 1e-7
    S.dev.close()


def test_baseline_config2_3d_single_cell_r1_apply(hip_lib):
    """BASELINE configs[1] at r=1: 3D idealized single cell (124 416 tets), P1, Na/K/Cl + potential, synthetic seeded
    inputs: matrix-free applies vs the assembled reference operator."""
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(1, n_axons=1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    x = synthetic_state(pb)
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.update_dnphi()
    Aemi, _, _ = ko.assemble_emi(pb, want_B=False)
    dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < 1e-11
    dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
    y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
    for k in range(pb.N_ions):
        assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < 1e-11
    dev.close()


def test_rccl_code_path_with_one_rank(hip_lib):
    """RCCL refuses two ranks on one device, so the collective code path is exercised with ONE rank: KNP_FORCE_COMM=1 makes
    bench.py build the process group, the slab partition and the RCCL communicator, and every reduction / restricted
    residual goes through ncclAllReduce on the solver's stream.  Must reproduce the single-process iteration counts."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["bench.py", "--gpus", "1", "--resolution", "0", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ)
    plain = subprocess.run([sys.executable] + args, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0, plain.stderr[-2000:]
    env["KNP_FORCE_COMM"] = "1"
    forced = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                             "127.0.0.1", "--master-port", "29517"] + args, cwd=root, env=env, capture_output=True, text=True,
                            timeout=600)
    assert forced.returncode == 0, forced.stderr[-2000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in forced.stdout.splitlines() if l.startswith("{")][-1])
    assert a["config"]["emi_iters_per_step"] == b["config"]["emi_iters_per_step"]
    assert a["config"]["knp_iters_per_step"] == b["config"]["knp_iters_per_step"]


@pytest.mark.parametrize("which", ["hh_emix", "glial"])
def test_emix_device_ode_models_match_host(hip_lib, which):
    """Device implementations of the EMIx membrane models (cm / ms / mV; reference: examples/emix-simulations/mm_hh.py,
    mm_glial.py) against the host batch integrator running the vectorised Python right-hand sides."""
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace, FacetFunction
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh_emix, mm_glial
    ode = mm_hh_emix if which == "hh_emix" else mm_glial
    m, s, f = make_mesh_2D(1)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    dev = device_for(pb)
    Q = FacetSpace(m)
    rng = np.random.default_rng(5)
    K_e = 3.3 * (1 + 0.05 * rng.uniform(-1, 1, Q.dim()))
    models = []
    for on_dev in (False, True):
        mm = MembraneModel(ode, facet_f=f, tag=1, V=Q)
        mm.set_parameter_values({'Cm': lambda x: 2.0})
        if on_dev:
            assert mm.attach_device(dev)
        for name, val in (('K_e', K_e), ('Na_i', np.full(Q.dim(), 12.8)), ('E_K', np.full(Q.dim(), -93.6)),
                          ('E_Na', np.full(Q.dim(), 53.3))):
            mm.set_parameter(name, FacetFunction(Q, val))
        models.append(mm)
    for k in range(5):
        for mm in models:
            mm.step_lsoda(dt=0.1, stimulus={'stim_amplitude': 5.0}, stimulus_locator=lambda x: x[0] < 20e-6)
    sh, sd = models[0].states, models[1].states
    ph, pd = models[0].parameters, models[1].parameters
    assert np.abs(sh - sd).max() < 1e-8 * np.abs(sh).max()
    assert np.abs(ph[:, 8:10] - pd[:, 8:10]).max() < 1e-7 * np.abs(ph[:, 8:10]).max()
    dev.close()


def test_baseline_config5_emix_mesh(hip_lib):
    """BASELINE configs[4]: the reference's bundled EMIx reconstruction (121 617 tets, unstructured; read with
    knpemidg.h5lite) with glial + neuronal membranes, three splitting steps on the general (coordinate-path) kernels:
    the solves converge at the reference tolerances, the bulk stays electroneutral and the resting state persists away from
    the stimulated region."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "emix_simulations"))
    from emix_common import make_solver, solver_parameters, Constant
    S = make_solver()
    assert S.mesh.num_cells() == 121617 and S.mesh.num_vertices() == 22419
    assert S.dev.n_geometry_classes == 0
    assert all(mm['ode'].on_device for mm in S.mem_models)
    t = Constant(0.0)
    S.solve_system_active(0.3, t, solver_parameters(), filename=None, save_fields=False)
    assert len(S.emi_niter) == 3 and max(S.emi_niter) < 100 and max(max(k) for k in S.knp_niter) < 100
    c = S.c.array().reshape(2, -1)
    na = S.ion_list[-1]['c'].array().ravel()
    assert np.abs(c[0] - c[1] + na).max() < 1e-9 * np.abs(c[1]).max()          # z = +1, -1, +1 and rho = 0
    phiM = S.phi_M_prev_PDE.array()
    glial = S.mem_models[0]['ode'].indices
    assert np.abs(phiM[glial] + 83.085).max() < 1.0                             # mV: glial rest potential
    assert np.isfinite(S.phi.array()).all()


def test_rho_sub_and_subdomain_diffusion_two_steps_vs_oracle(hip_lib):
    """The one shipped configuration with a non-zero background charge: run_tortuosity.py (rho_sub = -(Na + K - Cl) per subdomain
    :116-121, D_k / lambda_sub^2 :154-156, ion_list = [K, Na, Cl] with Cl, z = -1, eliminated :229) through the `Solver` API on a
    17 920-tet piece of the tissue mesh (glial + neuronal membranes): two full splitting steps against the oracle's assembled forms
    and direct solves fed with the same membrane outputs; rho enters the eliminated concentration (solver.py:831-838) and through
    it kappa and the KNP right-hand side of the second step."""
    from collections import namedtuple
    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "emix_simulations")
    if ex not in sys.path:
        sys.path.insert(0, ex)
    import emix_common as E
    from emix_sub import emix_submesh
    from knpemidg.models import mm_glial, mm_hh_emix
    P = ko.tortuosity_params()
    C = E.Constant
    params = namedtuple('params', ('dt', 'n_steps_ODE', 'F', 'psi', 'C_phi', 'C_M', 'R', 'temperature', 'phi_M_init_type', 'rho_sub'))(
        P["dt"], 25, P["F"], P["F"] / (P["R"] * P["temperature"]), P["C_phi"], P["C_M"], P["R"], P["temperature"], 'constant',
        {s: C(P["rho"][s]) for s in range(3)})

    def ion(name):
        return {'c_init_sub': {s: C(P["init"][name][s]) for s in range(3)}, 'c_init_sub_type': 'constant', 'bdry': C(0),
                'z': P["z"][name], 'name': name, 'D_sub': {s: C(P["D"][name] / P["lam"][s] ** 2) for s in range(3)}, 'f_source': C(0)}
    ion_list = [ion('K'), ion('Na'), ion('Cl')]
    stim = namedtuple('membrane_params', ('g_syn_bar', 'stimulus', 'stimulus_locator'))(5, {'stim_amplitude': 5}, lambda x: (x[0] < 3.0e-4))
    mt = emix_submesh()
    S = E.SolverEMIx(params, ion_list)
    S.verbose = False
    S.setup_domain(*mt)
    S.setup_parameters()
    S.setup_FEM_spaces()
    S.setup_membrane_model(stim, {1: mm_glial, 2: mm_hh_emix})
    S._unpack_solver_params(E.solver_parameters()._replace(rtol_emi=1e-11, rtol_knp=1e-13))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    pb = ko.build_tortuosity(mt[0], mt[1].array(), mt[2].array())
    assert np.abs(pb.rho).min() > 10.0 and len(np.unique(pb.ions[0]["D"])) == 2
    assert relerr(S.c.array(), pb.c) < 1e-15 and relerr(S.ion_list[-1]['c'].array(), pb.c_elim) < 1e-15
    vol = pb.geom.vol
    t = E.Constant(0.0)
    for k in range(2):
        S.step_membrane_models(k)
        pb.phi_M = S.phi_M_prev_PDE.array().copy()
        for name in pb.I_ch:
            pb.I_ch[name] = np.zeros(pb.mesh.num_facets())
            for mm in S.mem_models:
                a = mm['I_ch_k'][name].array()
                pb.I_ch[name][mm['ode'].indices] = a[mm['ode'].indices]
        S.solve_for_time_step(k, t)
        Eo = ko.solve_for_time_step(pb, direct=True)
        assert relerr(mean_free(S.phi.array(), vol), mean_free(pb.phi, vol)) < 1e-6
        assert relerr(S.c.array(), pb.c) < 1e-8
        assert relerr(S.ion_list[-1]['c'].array(), pb.c_elim) < 1e-8
        assert relerr(S.phi_M_prev_PDE.array()[pb.mem], pb.phi_M[pb.mem]) < 1e-6
        for ion_ in S.ion_list:
            assert relerr(ion_['E'].array()[pb.mem], Eo[ion_['name']]) < 1e-7
    # the background charge is what keeps the bulk electroneutral: z = (+1, +1, -1), rho != 0
    c = S.c.array().reshape(2, -1, 4)
    cl = S.ion_list[-1]['c'].array().reshape(-1, 4)
    assert np.abs(c[0] + c[1] - cl + pb.rho[:, None]).max() < 1e-9 * np.abs(cl).max()
    S.dev.close()


def test_solver_emi_variant(hip_lib):
    """`SolverEMI` (reference: src/knpemidg/solver_emi.py): potential-only stepping with frozen concentrations, HH membrane
    ODEs in the loop; every step's phi and phi_M against the oracle's EMI solve fed with the same ODE outputs."""
    from common_examples import physical_setup, solver_parameters, Constant
    from knpemidg import SolverEMI
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.models import mm_hh
    mesh, sub, surf = make_mesh_2D(0)
    params, ion_list, stim = physical_setup(1.0e-4)
    S = SolverEMI(params, ion_list)
    S.verbose = False
    S.setup_domain(mesh, sub, surf)
    S.setup_parameters()
    S.setup_FEM_spaces()
    S.setup_membrane_model(stim, {1: mm_hh})
    sp = solver_parameters(2, 0)._replace(rtol_emi=1e-11)
    S._unpack_solver_params(sp)
    S.splitting_scheme = True
    S.save_fields = S.save_solver_stats = False
    S.setup_varform_emi(); S.setup_solver_emi()
    pb = ko.build_idealized(mesh, sub.array(), surf.array(), membrane_tags=(1,))
    c0 = S.c.array().copy()
    t = Constant(0.0)
    for k in range(3):
        S.step_membrane_models(k)
        pb.phi_M = S.phi_M_prev_PDE.array().copy()
        for ion in pb.ions:
            pb.I_ch[ion["name"]] = S.mem_models[0]['I_ch_k'][ion["name"]].array().copy()
        S.solve_for_time_step(k, t)
        ko.solve_emi(pb, direct=True)
        ko.update_phi_M(pb)
        vol = pb.geom.vol
        assert relerr(mean_free(S.phi.array(), vol), mean_free(pb.phi, vol)) < 1e-6
        assert relerr(S.phi_M_prev_PDE.array()[pb.mem], pb.phi_M[pb.mem]) < 1e-6
    assert np.array_equal(S.c.array(), c0)                       # concentrations frozen
    assert abs(float(t) - 3e-4) < 1e-12


@pytest.mark.parametrize("shared", ["1", "0"])
def test_knp_hierarchy_shared_or_per_species(hip_lib, monkeypatch, shared):
    """KNP preconditioner variants: one shared hierarchy carrying the species as V-cycle columns (default when the
    diffusion coefficients are close) vs one hierarchy per species on concurrent streams.  Both must converge to the
    oracle's step (the preconditioner does not change the converged solution)."""
    from common_examples import make_solver, solver_parameters, Constant
    monkeypatch.setenv("KNP_AMG_SHARED", shared)
    mt = small_3d((10, 4, 4))
    S = make_solver(dim=3, resolution=0, n_axons=1, mesh_tuple=mt)
    sp = solver_parameters(3, 0)._replace(rtol_emi=1e-10, rtol_knp=1e-12)
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    pb = ko.build_idealized(mt[0], mt[1].array(), mt[2].array(), membrane_tags=(1,))
    t = Constant(0.0)
    for k in range(2):
        S.step_membrane_models(k)
        pb.phi_M = S.phi_M_prev_PDE.array().copy()
        for ion in pb.ions:
            pb.I_ch[ion["name"]] = S.mem_models[0]['I_ch_k'][ion["name"]].array().copy()
        S.solve_for_time_step(k, t)
        ko.solve_for_time_step(pb, direct=True)
        assert relerr(S.c.array(), pb.c) < 1e-8
    assert max(max(n) for n in S.knp_niter) < 60          # the auxiliary space is active (block-Jacobi alone needs hundreds)


def test_setup_helper_processes_and_their_fallback(hip_lib):
    """knpemidg/setup_worker.py: the first EMI hierarchy and the KNP hierarchies come from the helper processes when the device state
    still is the initial state they were given (the conforming spaces are then never built in this process); a changed state discards
    the helper's EMI result and builds in-process.  Both ways the step matches the oracle."""
    from common_examples import make_solver, solver_parameters, Constant
    from knpemidg import _abi as A
    for changed in (False, True):
        mt = small_3d((10, 4, 4))
        S = make_solver(dim=3, resolution=0, n_axons=1, mesh_tuple=mt)
        assert S._emi_helper is not None and S._knp_helper is not None
        pb = ko.build_idealized(mt[0], mt[1].array(), mt[2].array(), membrane_tags=(1,))
        if changed:
            S.dev.upload(A.F_C, 1.2 * S._init_c)
            S.dev.upload(A.F_C_PREV, 1.2 * S._init_c)
            S.dev.upload(A.F_C_ELIM, 1.2 * S._init_c_elim)
            pb.c *= 1.2; pb.c_prev_n *= 1.2; pb.c_elim *= 1.2
        S._unpack_solver_params(solver_parameters(3, 0)._replace(rtol_emi=1e-10, rtol_knp=1e-12))
        S.save_fields = S.save_solver_stats = False
        S.splitting_scheme = True
        S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
        assert S._emi_helper is None and S._knp_helper is None
        assert hasattr(S, "_cspace") == changed                 # in-process build only when the helper's result was discarded
        t = Constant(0.0)
        S.step_membrane_models(0)
        pb.phi_M = S.phi_M_prev_PDE.array().copy()
        for ion in pb.ions:
            pb.I_ch[ion["name"]] = S.mem_models[0]['I_ch_k'][ion["name"]].array().copy()
        S.solve_for_time_step(0, t)
        ko.solve_for_time_step(pb, direct=True)
        assert relerr(S.c.array(), pb.c) < 1e-8
        assert max(S.emi_niter) < 40 and max(max(n) for n in S.knp_niter) < 60
        S.dev.close()


def test_amg_hierarchy_is_refreshed_when_kappa_drifts(hip_lib, monkeypatch):
    """The reference rebuilds its AMG preconditioner at every solve (solver.py:505); this build lags it and refreshes when the
    coefficient has drifted (Solver._maybe_refresh_amg_emi): after the concentrations are scaled by 1.6 mid-run the EMI
    hierarchy is rebuilt from the new kappa and the solves keep converging in a handful of iterations."""
    from common_examples import make_solver, solver_parameters, Constant
    monkeypatch.setenv("KNP_AMG_REFRESH_EVERY", "2")
    S = make_solver(dim=3, resolution=0, n_axons=1)
    S._unpack_solver_params(solver_parameters(3, 0))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = Constant(0.0)
    for k in range(4):
        S.step_membrane_models(k); S.solve_for_time_step(k, t)
    assert getattr(S, "amg_refreshes", 0) == 0
    before = max(S.emi_niter)
    for f in (S.c, S.c_prev_n, S.ion_list[-1]['c']):
        f.set(1.6 * f.array())
    for k in range(4, 10):
        S.step_membrane_models(k); S.solve_for_time_step(k, t)
    assert S.amg_refreshes >= 1
    assert max(S.emi_niter[-3:]) <= 2 * before + 4, S.emi_niter
    S.dev.close()


def test_result_file_has_the_reference_datasets(hip_lib, tmp_path):
    """solve_system_active(..., save_fields=True) writes `<filename>results.h5` with the reference's dataset names
    (solver.py:1214-1242): mesh, subdomains, surfaces and one vector_n per saved step for concentrations, eliminated
    concentration and potential; the last snapshot equals the solver's final fields."""
    from common_examples import make_solver, solver_parameters, Constant
    from knpemidg.h5lite import H5File
    S = make_solver(dim=2, resolution=0)
    t = Constant(0.0)
    prefix = str(tmp_path / "out") + "/"
    S.solve_system_active(3e-4, t, solver_parameters(2, 0), filename=prefix, save_fields=True, save_solver_stats=True)
    f = H5File(prefix + "results.h5")
    for name in ("mesh/coordinates", "mesh/topology", "subdomains/values", "surfaces/values", "concentrations/vector_0",
                 "concentrations/vector_3", "elim_concentration/vector_3", "potential/vector_3", "potential/cell_dofs"):
        assert name in f.datasets, name
    assert "potential/vector_4" not in f.datasets
    assert np.array_equal(f.read("mesh/topology"), S.mesh.cells)
    assert np.array_equal(f.read("concentrations/vector_3"), S.c.array().ravel())
    assert np.array_equal(f.read("potential/vector_3"), S.phi.array().ravel())
    assert np.array_equal(f.read("surfaces/values"), S.surfaces.array())
    assert os.path.exists(prefix + "solver/emi_niter_0.txt")
    S.dev.close()


@pytest.mark.parametrize("dim,degree", [(2, 1), (3, 1), (2, 2)])
def test_f_source_array_and_callable_vs_oracle(hip_lib, dim, degree):
    """ion['f_source'] is any UFL coefficient in the reference (`L += ion['f_source'] * v_c * dx(0)`, solver.py:599): a Constant in
    the idealized examples, a box-and-time-window Expression in examples/local-astrocyte-depolarization/run_tortuosity.py:180-200.
    Here: a per-cell array for one species and a callable f(x, t) with exactly that shape for the other; L_knp on the device
    against the oracle's, inside and outside the time window, and the source's own contribution (difference to the source-free
    right-hand side) at the apply tolerance."""
    from idealized_common import make_solver, solver_parameters
    from knpemidg import _abi as A
    mt = None if dim == 2 else small_3d()
    S = make_solver(dim=dim, resolution=0, n_axons=1, degree=degree, mesh_tuple=mt)
    mesh = S.mesh
    lo, hi = mesh.coords.min(axis=0), mesh.coords.max(axis=0)
    a, b = lo + 0.2 * (hi - lo), lo + 0.7 * (hi - lo)
    g_syn, t0, t1 = 40.0, 2.0e-4, 6.0e-4

    def box(X, t):
        inside = np.all((X >= a) & (X <= b), axis=-1)
        return g_syn * inside * (t0 <= t) * (t <= t1) * (1.0 + 0.3 * np.sin(4.0e5 * X[..., 0]))
    rng = np.random.default_rng(8)
    per_cell = rng.uniform(-5.0, 5.0, mesh.num_cells())
    pb = ko.build_idealized(mesh, S.subdomains.array(), S.surfaces.array(), p=degree, membrane_tags=(1,))
    S.ion_list[0]['f_source'] = box
    S.ion_list[1]['f_source'] = per_cell
    S._unpack_solver_params(solver_parameters(dim, 0))
    S.splitting_scheme = True
    S.setup_varform_emi()
    S.step_membrane_models(0)                      # identical inputs on both sides: the oracle takes the device's state
    nf = mesh.num_facets()
    pb.c = S.c.array().reshape(pb.c.shape); pb.c_prev_n = S.c_prev_n.array().reshape(pb.c.shape)
    pb.c_elim = S.ion_list[-1]['c'].array().reshape(pb.c_elim.shape); pb.phi = S.phi.array().reshape(pb.phi.shape)
    pb.phi_M = S.phi_M_prev_PDE.array().copy()
    Ich = S.dev.download(A.F_I_CH).reshape(3, nf)
    for k, ion in enumerate(pb.ions):
        pb.I_ch[ion["name"]] = Ich[k].copy()
    base = [ko.knp_rhs(pb, k) for k in range(2)]
    pb.f_source = [box, per_cell]
    for t, active in ((0.0, False), (3.0e-4, True)):
        S._update_sources(t)
        pb.t = t
        S.dev.knp_rhs()
        got = S.dev.download(A.F_B_KNP).reshape(2, -1)
        for k in range(2):
            ref = ko.knp_rhs(pb, k).ravel()
            assert relerr(got[k], ref) < 1e-11
            d_ref = ref - base[k].ravel()
            assert (np.abs(d_ref).max() > 0) == (active or k == 1)
            if np.abs(d_ref).max() > 0:          # the source is 1e-3 ... 1e-5 of the mass term: 1e-11 of the total = 1e-6 of its own size
                assert relerr(got[k] - base[k].ravel(), d_ref) < 1e-6
    S.dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dim,degree", [(2, 1), (3, 1), (3, 2)])
def test_knp_load_measure_matches_host_formula(hip_lib, dim, degree, monkeypatch):
    """knp_knp_load_measure (the device pass that scales the EMI residual target, knpemidg/solver.py: _knp_load_norm): sums of
    (|b_K| / vol_K)^8 -- or |b_K|^2 / vol_K with KNP_KNP_NORM2=1 -- over the cells of the KNP right-hand side, against numpy on the
    downloaded field (1 / vol is stored in fp32 on the device: 1e-6)."""
    from common_examples import make_solver, solver_parameters
    from knpemidg.mesh import make_mesh_2D
    from knpemidg import _abi
    mt = make_mesh_2D(0) if dim == 2 else small_3d((6, 3, 3))
    S = make_solver(dim=dim, resolution=0, n_axons=1, mesh_tuple=mt, degree=degree)
    S._unpack_solver_params(solver_parameters(dim, 0))
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp()
    dev = S.dev
    dev.update_dnphi(); dev.knp_rhs()
    b = dev.download(_abi.F_B_KNP).reshape(S.N_ions, S.mesh.num_cells(), S.nd)
    x = S.mesh.coords[S.mesh.cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / (2.0 if dim == 2 else 6.0)
    bK2 = (b ** 2).sum(axis=2)
    want8 = ((bK2 / vol[None, :] ** 2) ** 4).sum(axis=1)
    got8 = dev.knp_load_measure()
    assert got8.shape == (S.N_ions,) and np.all(want8 > 0)
    assert np.abs(got8 / want8 - 1.0).max() < 1e-5                   # eight powers of the fp32 weight
    monkeypatch.setenv("KNP_KNP_NORM2", "1")
    got2 = dev.knp_load_measure()
    assert np.abs(got2 / (bK2 / vol[None, :]).sum(axis=1) - 1.0).max() < 1e-6
    monkeypatch.delenv("KNP_KNP_NORM2")
    # and the solver's target uses it
    z = np.abs([float(ion['z']) for ion in S.ion_list[:-1]])
    assert abs(S._knp_load_norm() / float(np.min(z * want8 ** 0.125)) - 1.0) < 1e-6
