"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.
Tolerances: operator applies / right-hand sides / projections are pure re-orderings of the same
floating-point sums -> 1e-11 relative to the vector's max norm (<= 1e-12 typical)."""
import numpy as np
import pytest

import knpemi_oracle as ko
from common import synthetic_state, device_for, push_state, relerr

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _problem(name):
    """One seeded-input problem by name (built on demand: each one costs the oracle's geometry and space tables)."""
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D
    from common import small_3d
    if name == "2D_neuron_r0":
        m, s, f = make_mesh_2D(0)
        return ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    if name == "3D_4axon_r0":
        m, s, f = make_mesh_3D(0)
        return ko.build_idealized(m, s.array(), f.array())
    if name == "3D_1axon_r0":
        m, s, f = make_mesh_3D(0, n_axons=1)
        return ko.build_idealized(m, s.array(), f.array())
    if name == "3D_1axon_r1":
        m, s, f = make_mesh_3D(1, n_axons=1)
        return ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    if name == "2D_neuron_r1_P2":       # DG-P2 (BASELINE configs[2] degree)
        m, s, f = make_mesh_2D(1)
        return ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    if name == "3D_box_P2":
        m, s, f = small_3d((8, 4, 4))
        return ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    # rho_sub != 0, D per subdomain, eliminated ion with z = -1 (run_tortuosity.py:116-121, 154-156, 229; solver.py:831-838)
    if name == "3D_4axon_r0_rho":
        from common import tortuosity_3d
        return tortuosity_3d(0)
    if name == "emix_sub_rho":
        import emix_sub
        m, s, f = emix_sub.emix_submesh()
        return ko.build_tortuosity(m, s.array(), f.array())
    raise KeyError(name)


@pytest.fixture(scope="module", params=["2D_neuron_r0", "3D_1axon_r0", "3D_4axon_r0", "2D_neuron_r1_P2", "3D_box_P2", "3D_4axon_r0_rho",
                                        "emix_sub_rho"])
def case(request, hip_lib):
    from knpemidg import _abi as A
    pb = _problem(request.param)
    x = synthetic_state(pb, volt=1.0e3 if request.param.endswith("_rho") else 1.0)      # the run_tortuosity.py units are cm / ms / mV
    dev = device_for(pb)
    push_state(dev, pb)
    yield pb, dev, x, A
    dev.close()


def test_kappa(case):
    pb, dev, x, A = case
    dev.update_kappa()
    assert relerr(dev.download(A.F_KAPPA), pb.kappa()) < 1e-14


def test_emi_apply(case):
    pb, dev, x, A = case
    Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
    dev.update_kappa()
    dev.upload(A.F_X, x[0])
    dev.emi_apply(A.F_X, A.F_Y)
    y = dev.download(A.F_Y, 0, pb.ndof)
    assert relerr(y, Aemi @ x[0].ravel()) < TOL
    # A is symmetric and annihilates constants
    dev.upload(A.F_X, np.ones(pb.ndof))
    dev.emi_apply(A.F_X, A.F_Y)
    y1 = dev.download(A.F_Y, 0, pb.ndof)
    assert np.abs(y1).max() < 1e-9 * np.abs(y).max()


def test_emi_rhs(case):
    pb, dev, x, A = case
    for splitting in (True, False):
        pb.splitting = splitting
        z = [ion["z"] for ion in pb.ions]
        D = np.stack([ion["D"] for ion in pb.ions])
        dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=splitting)
        dev.emi_rhs()
        assert relerr(dev.download(A.F_B_EMI), ko.emi_rhs(pb)) < TOL
    pb.splitting = True
    dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=True)


def test_knp_apply(case):
    pb, dev, x, A = case
    dev.update_dnphi()
    dev.upload(A.F_X, x)
    dev.knp_apply(A.F_X, A.F_Y)
    y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
    for k in range(pb.N_ions):
        Ak = ko.assemble_knp(pb, k)
        assert relerr(y[k], Ak @ x[k].ravel()) < TOL


_NORING = {"KNP_APPLY_RING": "0"}


@pytest.mark.parametrize("env,variant", [({}, 7), (_NORING, 6), ({"KNP_APPLY_MAT": "0"}, 2), (dict(_NORING, KNP_HALO_DYN="0"), 6),
                                         (dict(_NORING, KNP_HALO_NQ="64"), 6), (dict(_NORING, KNP_HALO_WG_PER_CU="1"), 6),
                                         ({"KNP_APPLY_HALO": "0"}, 1)])
def test_knp_apply_kernel_variants(hip_lib, monkeypatch, env, variant):
    """Every selectable KNP apply kernel of the structured 3D P1 path against the oracle matrix: the ring-staged kernel (loader wave +
    LDS-DMA ring, default), the halo-staged persistent kernel with the material table, with per-cell D, with strided instead of drawn
    blocks, other queue counts / one workgroup per CU (many blocks per workgroup), and the LDS-staged kernel of round 1."""
    from knpemidg import _abi as A
    pb = _problem("3D_4axon_r0")
    x = synthetic_state(pb)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(1) == variant
        dev.update_dnphi()
        dev.upload(A.F_X, x)
        dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


@pytest.mark.parametrize("env,evariant", [({}, 3), ({"KNP_EMI_RING": "0"}, 1)])
@pytest.mark.parametrize("which", ["3D_4axon_r0", "3D_1axon_r0"])
def test_emi_apply_kernel_variants(hip_lib, monkeypatch, env, evariant, which):
    """Both EMI apply kernels of the structured 3D P1 path (ring-staged: default; LDS-staged thread-per-cell kernel) against the
    oracle matrix, on meshes with membrane facets of one and of four cells."""
    from knpemidg import _abi as A
    pb = _problem(which)
    x = synthetic_state(pb)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(0) == evariant
        dev.update_kappa()
        dev.upload(A.F_X, x[0])
        dev.emi_apply(A.F_X, A.F_Y)
        Aemi, _, _ = ko.assemble_emi(pb, want_B=False)
        assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < TOL
    finally:
        dev.close()


_STEADY = {}


def _steady_case(which):
    """Problem, seeded inputs and the oracle products A x (assembled once per mesh: the r=1 matrices take the longest)."""
    if which not in _STEADY:
        pb = _problem(which)
        x = synthetic_state(pb)
        Aemi, _, _ = ko.assemble_emi(pb, want_B=False)
        ye = Aemi @ x[0].ravel()
        del Aemi
        yk = [ko.assemble_knp(pb, k) @ x[k].ravel() for k in range(pb.N_ions)]
        _STEADY[which] = (pb, x, ye, yk)
    return _STEADY[which]


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("which,wg", [("3D_4axon_r0", 8), ("3D_4axon_r0", 16), ("3D_1axon_r1", 8), ("3D_1axon_r1", 24)])
def test_ring_kernels_steady_state_vs_oracle(hip_lib, monkeypatch, which, wg, split):
    """The ring-staged applies in the regime the benchmarked r=2 run spends its time in: KNP_RING_WG shrinks the launch so that one
    workgroup walks MANY blocks (r=0 4-axon mesh: 61 blocks on 8 / 16 workgroups = 7-8 / 3-4 each; single-axon r=1 mesh: 486 blocks on
    8 / 24 workgroups = 61 / 20-21 each) -- slot reuse in the three- / four-slot rings, the wrap of the four list buffers, the counted
    s_waitcnt vmcnt(N) that leaves block n + 2 in flight across the barrier, and the short last block -- against the oracle's assembled
    CSR operators (src/knpemidg/solver.py:325-328, 586-594) at 1e-11; KNP with one consumer group and species-split groups."""
    from knpemidg import _abi as A
    pb, x, ye, yk = _steady_case(which)
    monkeypatch.setenv("KNP_RING_WG", str(wg))
    monkeypatch.setenv("KNP_RING_SPLIT", split)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(0) == 3 and dev.apply_variant(1) == 7
        dev.update_kappa(); dev.update_dnphi()
        if split == "0":
            dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
            assert relerr(dev.download(A.F_Y, 0, pb.ndof), ye) < TOL
        dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], yk[k]) < TOL
    finally:
        dev.close()


def test_ring_kernel_one_species_steady_state_vs_oracle(hip_lib, monkeypatch):
    """NS = 1 instance of the ring-staged KNP apply with 7-8 blocks per workgroup against the oracle."""
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(0)
    P = ko.idealized_params()
    nc = m.num_cells()
    ions = [dict(name=n, z=P["z"][n], D=np.full(nc, P["D"][n])) for n in ("K", "Cl")]
    pb = ko.Problem(m, s.array().astype(np.int64), f.array(), 1, ions, P, membrane_tags=(1, 2))
    rng = np.random.default_rng(11)
    pb.c = rng.uniform(50.0, 150.0, size=pb.c.shape)
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = rng.uniform(50.0, 150.0, size=pb.c_elim.shape)
    x = synthetic_state(pb)
    monkeypatch.setenv("KNP_RING_WG", "8")
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(1) == 7
        dev.update_dnphi()
        dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


@pytest.mark.parametrize("names,variant", [(("K", "Cl"), 7), (("K", "Cl", "Na", "Ca"), 1)])
def test_knp_apply_with_one_and_three_solved_species(hip_lib, names, variant):
    """Species counts other than the reference's two solved ions: ONE solved species runs the ring-staged kernel's NS = 1 instance,
    THREE run the LDS-staged kernel (the ring- and halo-staged ones carry at most two); same operators as the oracle's."""
    from knpemidg import _abi as A
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(0)
    P = ko.idealized_params()
    nc = m.num_cells()
    z = dict(P["z"], Ca=2.0)
    Dc = dict(P["D"], Ca=0.8e-9)
    ions = [dict(name=n, z=z[n], D=np.full(nc, Dc[n])) for n in names]
    pb = ko.Problem(m, s.array().astype(np.int64), f.array(), 1, ions, P, membrane_tags=(1, 2))
    rng = np.random.default_rng(11)
    pb.c = rng.uniform(50.0, 150.0, size=pb.c.shape)
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = rng.uniform(50.0, 150.0, size=pb.c_elim.shape)
    x = synthetic_state(pb)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(1) == variant
        dev.update_dnphi()
        dev.upload(A.F_X, x)
        dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


def test_knp_apply_with_cellwise_diffusion(hip_lib):
    """D that differs from cell to cell (more distinct coefficient tuples than the material table holds): the halo-staged kernel
    stages D itself; same operator as the oracle's."""
    from knpemidg import _abi as A
    pb = _problem("3D_4axon_r0")
    rng = np.random.default_rng(5)
    for ion in pb.ions:
        ion["D"] = np.asarray(ion["D"], dtype=float) * rng.uniform(0.5, 1.5, size=len(pb.cell_tags))
    x = synthetic_state(pb)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.apply_variant(1) == 2
        dev.update_dnphi()
        dev.upload(A.F_X, x)
        dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


def test_knp_rhs(case):
    pb, dev, x, A = case
    dev.knp_rhs()
    b = dev.download(A.F_B_KNP).reshape(pb.N_ions, -1)
    for k in range(pb.N_ions):
        assert relerr(b[k], ko.knp_rhs(pb, k)) < TOL


def test_rhs_without_splitting(case):
    """Original (non-splitting) Robin data: g = phi_M - I_ch / C_phi in L_emi (solver.py:337) and
    g_k = phi_M - dt / (C_M alpha) I_ch_k in L_knp without the + dt / C_M I_ch term (solver.py:619-622)."""
    pb, dev, x, A = case
    z = [ion["z"] for ion in pb.ions]
    D = np.stack([ion["D"] for ion in pb.ions])
    try:
        pb.splitting = False
        dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=False)
        dev.emi_rhs()
        ref_split = None
        assert relerr(dev.download(A.F_B_EMI), ko.emi_rhs(pb)) < TOL
        dev.knp_rhs()
        b = dev.download(A.F_B_KNP).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            ref = ko.knp_rhs(pb, k)
            assert relerr(b[k], ref) < TOL
            pb.splitting = True
            ref_split = ko.knp_rhs(pb, k)
            pb.splitting = False
            assert relerr(ref, ref_split) > 1e-10         # the two Robin data really differ on this state
    finally:
        pb.splitting = True
        dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=True)


def test_step_updates(case):
    pb, dev, x, A = case
    import copy
    dev.step_updates()
    phiM = dev.download(A.F_PHI_M)
    ref = ko.update_phi_M(pb).copy()
    assert relerr(phiM[pb.mem], ref) < 1e-13
    celim = dev.download(A.F_C_ELIM)
    assert relerr(celim, ko.update_c_elim(pb)) < 1e-14
    E = dev.download(A.F_E).reshape(len(pb.ions), -1)
    for k in range(len(pb.ions)):
        assert relerr(E[k][pb.mem], ko.nernst(pb, k)) < 1e-12
    assert relerr(dev.download(A.F_C_PREV), pb.c) < 1e-16
    # traces used by the update_ode hook (run_3D.py:44-49)
    K_e = dev.facet_trace(A.F_C, 0, 0)
    ref = ko.facet_average(pb, pb.mem, lambda plus, minus: plus(pb.c[0]), pb.p)
    assert relerr(K_e[pb.mem], ref) < 1e-14
    Na_i = dev.facet_trace(A.F_C_ELIM, 0, 1)
    ref = ko.facet_average(pb, pb.mem, lambda plus, minus: minus(pb.c_elim), pb.p)
    assert relerr(Na_i[pb.mem], ref) < 1e-14


@pytest.mark.parametrize("name", ["idealized_2D_r0", "box_3D_8x4x4", "box_3D_6x3x3_P2", "idealized_2D_r0_P2"])
def test_gpu_matches_golden(hip_lib, name):
    """HIP path vs the committed fixtures (tests/golden/*.npz): applies, right-hand sides and one converged
    splitting step.  The 2D r=0 case has membrane-tagged facets between EQUAL-tag cells (the mesh is too
    coarse for the ICS box): the reference's n('-') orientation rule (utils.py:80) is exercised."""
    import os
    from knpemidg import _abi as A
    from knpemidg.mesh import Mesh
    from common import mean_free
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    mesh = Mesh(g["coords"], g["cells"])
    pb = ko.build_idealized(mesh, g["cell_tags"], g["facet_tags"], p=int(g["degree"]) if "degree" in g else 1, membrane_tags=(1,))
    pb.c, pb.c_prev_n, pb.c_elim, pb.phi, pb.phi_M = g["c"], g["c_prev"], g["c_elim"], g["phi"], g["phi_M"]
    for k, ion in enumerate(pb.ions):
        pb.I_ch[ion["name"]] = g["I_ch"][k]
    dev = device_for(pb)
    push_state(dev, pb)
    dev.update_kappa(); dev.update_dnphi()
    dev.upload(A.F_X, g["x"][0]); dev.emi_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y, 0, pb.ndof), g["emi_Ax"]) < TOL
    dev.upload(A.F_X, g["x"]); dev.knp_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y), g["knp_Ax"]) < TOL
    dev.emi_rhs(); dev.knp_rhs()
    assert relerr(dev.download(A.F_B_EMI), g["emi_rhs"]) < TOL
    assert relerr(dev.download(A.F_B_KNP), g["knp_rhs"]) < TOL
    # one splitting step, tight tolerances
    dev.emi_solve(1e-11, maxit=50000)
    dev.update_dnphi(); dev.knp_rhs(); dev.knp_solve(1e-13, maxit=5000)
    dev.step_updates()
    vol = pb.geom.vol
    assert relerr(mean_free(dev.download(A.F_PHI), vol), mean_free(g["step_phi"], vol)) < 1e-6
    assert relerr(dev.download(A.F_C), g["step_c"]) < 1e-8
    assert relerr(dev.download(A.F_C_ELIM), g["step_c_elim"]) < 1e-8
    assert relerr(dev.download(A.F_PHI_M)[pb.mem], g["step_phi_M"][pb.mem]) < 1e-6
    assert relerr(dev.download(A.F_E).reshape(3, -1)[:, pb.mem], g["step_E"]) < 1e-7
    dev.close()


@pytest.mark.parametrize("world,p", [(2, 1), (3, 1), (2, 2)])
def test_partitioned_kernels_on_one_gpu(hip_lib, monkeypatch, world, p):
    """Owned+ghost sub-meshes on the device: every rank's context (all living on this one GPU, ghosts filled
    from the global arrays = what the RCCL halo exchange delivers) must reproduce the owned rows of the
    global oracle results.  Exercises nc_owned < nc in every kernel; RCCL itself needs >= 2 GPUs."""
    from knpemidg import _abi as A
    from knpemidg.partition import Partition
    from common import small_3d
    m, s, f = small_3d((12, 4, 4) if p == 1 else (8, 3, 3))
    pbg = ko.build_idealized(m, s.array(), f.array(), p=p, membrane_tags=(1,))
    x = synthetic_state(pbg)
    Ag, bg, _ = ko.assemble_emi(pbg, want_B=False)
    yg = (Ag @ x[0].ravel()).reshape(-1, pbg.nd)
    yk = np.stack([(ko.assemble_knp(pbg, k) @ x[k].ravel()).reshape(-1, pbg.nd) for k in range(pbg.N_ions)])
    bk = np.stack([ko.knp_rhs(pbg, k).reshape(-1, pbg.nd) for k in range(pbg.N_ions)])
    import copy
    q = copy.deepcopy(pbg)
    ko.update_phi_M(q); ko.update_c_elim(q)
    Eg = np.stack([ko.nernst(q, k) for k in range(3)])
    part = Partition(m, world, method="slab" if world == 2 else "rcb")
    for rank in range(world):
        loc = part.local(rank)
        sub_l, surf_l = loc.localize(s, f, (1,))
        pbl = ko.build_idealized(loc.mesh, sub_l.array(), surf_l.array(), p=p, membrane_tags=(1,))
        cg, no = loc.cells_global, loc.nc_owned
        pbl.c, pbl.c_prev_n, pbl.c_elim, pbl.phi = pbg.c[:, cg], pbg.c_prev_n[:, cg], pbg.c_elim[cg], pbg.phi[cg]
        pbl.phi_M = pbg.phi_M[loc.facets_global]
        for name in pbg.I_ch:
            pbl.I_ch[name] = pbg.I_ch[name][loc.facets_global]
        dev = device_for(pbl, nc_owned=no)
        assert 0 < dev.n_interior < no                        # device order: interior cells first, cells on the cut after them
        push_state(dev, pbl)
        dev.update_kappa(); dev.update_dnphi()
        for split in ("0", "1"):                              # one launch / interior + boundary launches (the overlapped form)
            monkeypatch.setenv("KNP_FORCE_SPLIT", split)
            dev.upload(A.F_Y, np.zeros(dev.size(A.F_Y)))
            dev.upload(A.F_X, x[0][cg]); dev.emi_apply(A.F_X, A.F_Y)
            y = dev.download(A.F_Y, 0, pbl.ndof).reshape(-1, pbl.nd)
            assert relerr(y[:no], yg[cg[:no]]) < TOL
            dev.upload(A.F_X, x[:, cg]); dev.knp_apply(A.F_X, A.F_Y)
            y = dev.download(A.F_Y).reshape(pbg.N_ions, -1, pbl.nd)
            assert relerr(y[:, :no], yk[:, cg[:no]]) < TOL
        monkeypatch.delenv("KNP_FORCE_SPLIT")
        dev.emi_rhs(); dev.knp_rhs()
        assert relerr(dev.download(A.F_B_EMI).reshape(-1, pbl.nd)[:no], bg.reshape(-1, pbg.nd)[cg[:no]]) < TOL
        assert relerr(dev.download(A.F_B_KNP).reshape(pbg.N_ions, -1, pbl.nd)[:, :no], bk[:, cg[:no]]) < TOL
        dev.step_updates()
        lmem = pbl.mem                                       # local membrane facets touching an owned cell
        gmem = loc.facets_global[lmem]
        pos = np.searchsorted(pbg.mem, gmem)
        assert relerr(dev.download(A.F_PHI_M)[lmem], q.phi_M[gmem]) < 1e-13
        assert relerr(dev.download(A.F_E).reshape(3, -1)[:, lmem], Eg[:, pos]) < 1e-12
        assert relerr(dev.download(A.F_C_ELIM).reshape(-1, pbl.nd), q.c_elim[cg]) < 1e-14
        dev.close()


def test_emix_mesh_vs_oracle(hip_lib):
    """BASELINE configs[4] on its REAL mesh against the oracle: the reference's bundled tissue reconstruction (121 617 unstructured
    tets, three subdomain classes, slivers, glial + neuronal membrane tags, membranes between equal-tag cells; parameters of
    examples/emix-simulations/run_EMIx_simulation.py:56-170 in cm / ms / mV) -- both operator applies against the oracle's assembled
    CSR matrices, both right-hand sides in both splitting modes, the step-III projections and the update_ode traces, on seeded
    inputs.  No geometry classes on this mesh: the applies run the unstructured ring-staged kernels (apply_ring_u.hip)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "emix_simulations"))
    from emix_common import load_mesh
    from knpemidg import _abi as A
    m, s, f = load_mesh()
    pb = ko.build_emix(m, s.array(), f.array())
    assert m.num_cells() == 121617 and len(np.unique(pb.cell_tags)) == 3 and set(np.unique(pb.facet_tags[pb.mem])) == {1, 2}
    x = synthetic_state(pb, volt=1.0e3)
    dev = device_for(pb)
    try:
        assert dev.n_geometry_classes == 0
        assert dev.apply_variant(0) == 10 and dev.apply_variant(1) == 10       # ring-staged, geometry from staged vertex coordinates (apply_ring_u.hip)
        push_state(dev, pb)
        dev.update_kappa(); dev.update_dnphi()
        assert relerr(dev.download(A.F_KAPPA), pb.kappa()) < 1e-14
        Aemi, _, _ = ko.assemble_emi(pb, want_B=False)
        dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
        assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < TOL
        del Aemi
        dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
        z = [ion["z"] for ion in pb.ions]
        D = np.stack([ion["D"] for ion in pb.ions])
        for splitting in (False, True):
            pb.splitting = splitting
            dev.set_params(pb.C_M, pb.dt, pb.F, pb.R, pb.T, pb.C_phi, pb.tau, pb.tau, z, D, rho=pb.rho, splitting=splitting)
            dev.emi_rhs(); dev.knp_rhs()
            assert relerr(dev.download(A.F_B_EMI), ko.emi_rhs(pb)) < TOL
            b = dev.download(A.F_B_KNP).reshape(pb.N_ions, -1)
            for k in range(pb.N_ions):
                assert relerr(b[k], ko.knp_rhs(pb, k)) < TOL
        K_e = dev.facet_trace(A.F_C, 0, 0)
        assert relerr(K_e[pb.mem], ko.facet_average(pb, pb.mem, lambda plus, minus: plus(pb.c[0]), pb.p)) < 1e-14
        Na_i = dev.facet_trace(A.F_C_ELIM, 0, 1)
        assert relerr(Na_i[pb.mem], ko.facet_average(pb, pb.mem, lambda plus, minus: minus(pb.c_elim), pb.p)) < 1e-14
        dev.step_updates()
        assert relerr(dev.download(A.F_PHI_M)[pb.mem], ko.update_phi_M(pb).copy()) < 1e-13
        assert relerr(dev.download(A.F_C_ELIM), ko.update_c_elim(pb)) < 1e-14
        E = dev.download(A.F_E).reshape(len(pb.ions), -1)
        for k in range(len(pb.ions)):
            assert relerr(E[k][pb.mem], ko.nernst(pb, k)) < 1e-12
    finally:
        dev.close()


@pytest.mark.parametrize("env,variant", [({}, 10), ({"KNP_RING_WG": "8"}, 10), ({"KNP_APPLY_RING_U": "0"}, 0)])
@pytest.mark.parametrize("which", ["emix_sub", "emix_sub_refined", "jittered_box"])
def test_unstructured_apply_kernel_variants(hip_lib, monkeypatch, env, variant, which):
    """Both apply paths for 3D P1 meshes WITHOUT geometry classes against the oracle's assembled operators (solver.py:325-328, 586-594):
    the ring-staged kernels that form the geometry from staged vertex coordinates (apply_ring_u.hip; default), the same kernels with
    the launch shrunk to 8 workgroups (KNP_RING_WG: a workgroup walks 9-70 blocks -- both slots and both list buffers reused many
    times, the short last block) and the thread-per-cell coordinate kernels (KNP_APPLY_RING_U=0).  Meshes: a 17 920-tet piece of the
    tissue reconstruction (glial + neuronal membranes, slivers), its regular refinement (143 360 tets: the measured bench workload's
    shape) and a jittered box (cells of a structured mesh with every interior vertex moved)."""
    from knpemidg import _abi as A
    if which == "jittered_box":
        from common import small_3d
        m, s, f = small_3d((16, 6, 6))
        rng = np.random.default_rng(11)
        inner = np.ones(m.num_vertices(), dtype=bool)
        inner[np.unique(m.facets[m.exterior_facets()])] = False
        jit = rng.uniform(-1, 1, size=m.coords.shape) * np.array([0.2e-6, 0.02e-6, 0.02e-6])
        m.coords[inner] += jit[inner]
        pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
        volt = 1.0
    else:
        import emix_sub
        m, s, f = emix_sub.emix_submesh()
        if which == "emix_sub_refined":
            from knpemidg.mesh import refine_uniform, MeshFunction
            m, parent = refine_uniform(m)
            sub = s.array()[parent]
            fc = m.facet_cells
            tags = np.full(m.num_facets(), 10, dtype=np.uint32)
            it = fc[:, 1] >= 0
            s0, s1 = sub[fc[it, 0]], sub[fc[it, 1]]
            p0, p1 = parent[fc[it, 0]], parent[fc[it, 1]]
            # a child facet is a membrane facet iff it lies on a parent's membrane facet: the two children stem from different parents
            # whose shared facet carries that tag
            pf = emix_sub.emix_submesh()
            key = {}
            pm = pf[0]
            for fid in np.nonzero(pf[2].array() % 10 != 0)[0]:
                a, b = pm.facet_cells[fid]
                key[(min(a, b), max(a, b))] = int(pf[2].array()[fid])
            t = np.zeros(int(it.sum()), dtype=np.uint32)
            for q, (a, b) in enumerate(zip(np.minimum(p0, p1), np.maximum(p0, p1))):
                if a != b:
                    t[q] = key.get((int(a), int(b)), 0)
            tags[it] = t
            s, f = MeshFunction(m, 3, sub.astype(np.uint32)), MeshFunction(m, 2, tags)
        pb = ko.build_emix(m, s.array(), f.array())
        volt = 1.0e3
    x = synthetic_state(pb, volt=volt)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev = device_for(pb)
    try:
        assert dev.n_geometry_classes == 0
        assert dev.apply_variant(0) == variant and dev.apply_variant(1) == variant
        push_state(dev, pb)
        dev.update_kappa(); dev.update_dnphi()
        Aemi, _, _ = ko.assemble_emi(pb, want_B=False)
        dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
        assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < TOL
        del Aemi
        dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


@pytest.mark.parametrize("names,variant", [(("K", "Cl"), 10), (("K", "Cl", "Na", "Ca"), 0)])
def test_unstructured_knp_apply_with_one_and_three_solved_species(hip_lib, names, variant):
    """Species counts other than two on a mesh without geometry classes: ONE solved species runs the NS = 1 instance of the unstructured
    ring-staged kernel, THREE run the thread-per-cell coordinate kernel (the ring carries at most two); oracle operators at 1e-11."""
    from knpemidg import _abi as A
    import emix_sub
    m, s, f = emix_sub.emix_submesh()
    P = ko.emix_params()
    nc = m.num_cells()
    z = dict(P["z"], Ca=2.0)
    Dc = dict(P["D"], Ca=0.8e-8)
    tags = s.array().astype(np.int64)
    ions = [dict(name=n, z=z[n], D=np.full(nc, Dc[n]) * np.array([1.0, 0.5, 0.25])[tags]) for n in names]
    pb = ko.Problem(m, tags, f.array(), 1, ions, P, membrane_tags=(1, 2))
    rng = np.random.default_rng(11)
    pb.c = rng.uniform(50.0, 150.0, size=pb.c.shape)
    pb.c_prev_n = pb.c.copy()
    pb.c_elim = rng.uniform(50.0, 150.0, size=pb.c_elim.shape)
    x = synthetic_state(pb, volt=1.0e3)
    dev = device_for(pb)
    try:
        push_state(dev, pb)
        assert dev.n_geometry_classes == 0 and dev.apply_variant(1) == variant
        dev.update_dnphi()
        dev.upload(A.F_X, x)
        dev.knp_apply(A.F_X, A.F_Y)
        y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
        for k in range(pb.N_ions):
            assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    finally:
        dev.close()


def test_partitioned_emix_mesh_on_one_gpu(hip_lib, monkeypatch):
    """BASELINE configs[4] mesh (121 617 unstructured tets) cut into 4 parts by recursive coordinate bisection: a rank's owned +
    ghost context (ghost values as the halo exchange delivers them) reproduces the owned rows of the single-context applies,
    in one launch and as interior + boundary launches."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "emix_simulations"))
    from emix_common import load_mesh
    from knpemidg import _abi as A
    from knpemidg.partition import Partition
    m, s, f = load_mesh()
    rng = np.random.default_rng(5)
    nc = m.num_cells()
    D = np.full((3, nc), 1.5e-8)
    x = rng.uniform(-1, 1, size=(2, nc, 4))
    cc = 100.0 * (1 + 0.01 * rng.uniform(-1, 1, size=(2, nc, 4)))
    ce = 100.0 * (1 + 0.01 * rng.uniform(-1, 1, size=(nc, 4)))
    phi = 50.0 * rng.uniform(-1, 1, size=(nc, 4))

    def context(mesh, sub, surf, n_own, sel):
        dev = A.Device(mesh, sub, surf, (1, 2), 3, nc_owned=n_own)
        dev.set_params(2.0, 0.1, 96485e3, 8.314e3, 300e3, 20.0, 60.0, 60.0, [1.0, -1.0, 1.0], D[:, sel])
        dev.upload(A.F_C, cc[:, sel]); dev.upload(A.F_C_ELIM, ce[sel]); dev.upload(A.F_PHI, phi[sel])
        dev.update_kappa(); dev.update_dnphi()
        dev.upload(A.F_X, x[:, sel])
        dev.emi_apply(A.F_X, A.F_Y)
        ye = dev.download(A.F_Y, 0, len(sel) * 4).reshape(-1, 4)
        dev.knp_apply(A.F_X, A.F_Y)
        yk = dev.download(A.F_Y).reshape(2, -1, 4)
        return dev, ye, yk
    dev, ye, yk = context(m, s.array(), f.array(), None, np.arange(nc))
    dev.close()
    part = Partition(m, 4, method="rcb")
    for rank in (0, 3):
        loc = part.local(rank)
        sub_l, surf_l = loc.localize(s, f, (1, 2))
        cg, no = loc.cells_global, loc.nc_owned
        for split in ("0", "1"):
            monkeypatch.setenv("KNP_FORCE_SPLIT", split)
            dl, yel, ykl = context(loc.mesh, sub_l.array(), surf_l.array(), no, cg)
            assert 0 < dl.n_interior < no
            assert relerr(yel[:no], ye[cg[:no]]) < TOL and relerr(ykl[:, :no], yk[:, cg[:no]]) < TOL
            dl.close()
    monkeypatch.delenv("KNP_FORCE_SPLIT")


def test_unstructured_geometry_uses_coordinate_kernels(hip_lib):
    """Randomly jittered vertices destroy the translation invariance of the BoxMesh: no geometry classes can be
    formed, so the kernels that recompute the geometry from vertex coordinates run (ring-staged for the operator applies,
    thread-per-cell for the right-hand sides and the block inverses).  Also exercises non-uniform h, varying facet areas and
    cell volumes, and a converged solve on such a mesh."""
    from knpemidg import _abi as A
    from common import small_3d
    m, s, f = small_3d((10, 4, 4))
    rng = np.random.default_rng(11)
    interior_v = np.ones(m.num_vertices(), dtype=bool)
    interior_v[np.unique(m.facets[m.exterior_facets()])] = False
    jit = rng.uniform(-1, 1, size=m.coords.shape) * np.array([0.2e-6, 0.02e-6, 0.02e-6])
    m.coords[interior_v] += jit[interior_v]
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    x = synthetic_state(pb)
    dev = device_for(pb)
    assert dev.n_geometry_classes == 0
    push_state(dev, pb)
    dev.update_kappa(); dev.update_dnphi()
    Aemi, b, _ = ko.assemble_emi(pb, want_B=False)
    dev.upload(A.F_X, x[0]); dev.emi_apply(A.F_X, A.F_Y)
    assert relerr(dev.download(A.F_Y, 0, pb.ndof), Aemi @ x[0].ravel()) < TOL
    dev.upload(A.F_X, x); dev.knp_apply(A.F_X, A.F_Y)
    y = dev.download(A.F_Y).reshape(pb.N_ions, -1)
    for k in range(pb.N_ions):
        assert relerr(y[k], ko.assemble_knp(pb, k) @ x[k].ravel()) < TOL
    dev.emi_rhs(); dev.knp_rhs()
    assert relerr(dev.download(A.F_B_EMI), b) < TOL
    for k in range(pb.N_ions):
        assert relerr(dev.download(A.F_B_KNP).reshape(pb.N_ions, -1)[k], ko.knp_rhs(pb, k)) < TOL
    # a converged solve on the unstructured mesh (AMG built from the same conforming operator)
    from knpemidg import amg
    cs = amg.ConformingSpace(m, f.array(), (1,))
    dev.amg_upload(0, cs.dof, amg.build_hierarchy(cs.stiffness(pb.kappa(), membrane=(pb.mem, pb.C_phi))))
    dev.upload(A.F_PHI, np.zeros(pb.ndof))
    n, _ = dev.emi_solve(1e-10, maxit=5000)
    from common import mean_free
    ref = ko.solve_emi(pb, direct=True)
    assert relerr(mean_free(dev.download(A.F_PHI), pb.geom.vol), mean_free(ref, pb.geom.vol)) < 1e-6
    dev.close()


def test_full_size_properties_r2(hip_lib):
    """BASELINE configs[3] mesh (r=2: 995 328 tets, 11.9 M DoFs) is too large for the oracle, so the operators are
    checked through size-independent properties: EMI annihilates constants and is symmetric, both operators are
    linear, the KNP operator without drift conserves mass (1^T A_k 1 = |Omega| / dt), and one splitting step keeps
    electroneutrality and the rest state."""
    sys_path_examples()
    from idealized_common import make_solver, solver_parameters, Constant
    from knpemidg import _abi as A
    S = make_solver(dim=3, resolution=2)
    dev = S.dev
    assert dev.nc == 995328 and dev.n_geometry_classes > 0
    ndof = dev.nc * 4
    rng = np.random.default_rng(3)
    c_rest, ce_rest = dev.download(A.F_C), dev.download(A.F_C_ELIM)                        # restored before the rest-state step below
    dev.upload(A.F_C, c_rest * (1 + 0.01 * rng.uniform(-1, 1, size=c_rest.shape)))         # kappa varies inside every cell
    dev.upload(A.F_C_ELIM, ce_rest * (1 + 0.01 * rng.uniform(-1, 1, size=ce_rest.shape)))
    dev.update_kappa()
    x = rng.uniform(-1, 1, size=(2, ndof))
    pad = np.zeros(ndof)

    def emi(v):
        dev.upload(A.F_X, np.concatenate([v, pad])); dev.emi_apply(A.F_X, A.F_Y)
        return dev.download(A.F_Y, 0, ndof)
    y0, y1 = emi(x[0]), emi(x[1])
    assert np.abs(emi(np.ones(ndof))).max() < 1e-9 * np.abs(y0).max()                     # constants in the null space
    assert abs(x[1] @ y0 - x[0] @ y1) < 1e-10 * abs(x[1] @ y0)                             # symmetry
    assert relerr(emi(2.0 * x[0] - 0.5 * x[1]), 2.0 * y0 - 0.5 * y1) < 1e-12              # linearity
    # the ring-staged kernels in their steady state (15 of 15.2 rounds at this size: every slot and list buffer reused many times)
    # against the thread-per-cell LDS-staged kernels (oracle-verified, no persistence, no ring) on RANDOM vectors, a rough potential
    # (drift and upwinding active on every facet) and the perturbed kappa: the same floating-point sums up to their order
    import os
    phi_r = 0.07 * rng.uniform(-1, 1, size=ndof)
    dev.upload(A.F_PHI, phi_r); dev.update_dnphi()
    xk = rng.uniform(-1, 1, size=2 * ndof)

    def knp(v):
        dev.upload(A.F_X, v); dev.knp_apply(A.F_X, A.F_Y)
        return dev.download(A.F_Y)
    assert dev.apply_variant(0) == 3 and dev.apply_variant(1) == 7
    yk_ring = knp(xk)
    os.environ["KNP_RING_SPLIT"] = "1"
    try:
        yk_split = knp(xk)
    finally:
        del os.environ["KNP_RING_SPLIT"]
    os.environ["KNP_APPLY_HALO"] = "0"; os.environ["KNP_EMI_RING"] = "0"
    try:
        assert dev.apply_variant(0) == 1 and dev.apply_variant(1) == 1
        yk_staged = knp(xk)
        y0_staged = emi(x[0])
    finally:
        del os.environ["KNP_APPLY_HALO"], os.environ["KNP_EMI_RING"]
    assert np.abs(yk_staged).max() > 0
    assert relerr(yk_ring, yk_staged) < 1e-12 and relerr(yk_split, yk_staged) < 1e-12
    assert relerr(y0, y0_staged) < 1e-12
    del yk_ring, yk_split, yk_staged, y0_staged
    dev.upload(A.F_PHI, np.zeros(ndof)); dev.update_dnphi()                                # no drift
    dev.upload(A.F_X, np.ones(2 * ndof)); dev.knp_apply(A.F_X, A.F_Y)
    yk = dev.download(A.F_Y).reshape(2, -1)
    vol_total = 32e-6 * 0.9e-6 * 0.9e-6
    for k in range(2):
        assert abs(yk[k].sum() - vol_total / 1e-4) < 1e-9 * vol_total / 1e-4                # 1^T (M/dt + K_sipg) 1
    # one unstimulated splitting step from the calibrated rest state (run_3D.py:80-86)
    dev.upload(A.F_C, c_rest); dev.upload(A.F_C_ELIM, ce_rest); dev.update_kappa()
    S.stimulus = {}
    S._unpack_solver_params(solver_parameters(3, 2))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    c0 = S.c.array()
    t = Constant(0.0)
    S.step_membrane_models(0); S.solve_for_time_step(0, t)
    c1, ce = S.c.array(), S.ion_list[-1]['c'].array()
    # rest state stays at rest up to the drift driven by the EMI solver tolerance (rtol 1e-5 on phi, as in the reference)
    assert relerr(c1, c0) < 1e-5
    assert np.abs(c1[0] - c1[1] + ce).max() < 1e-9 * np.abs(ce).max()                       # z = (+1, -1, +1): electroneutral
    pm = S.phi_M_prev_PDE.array()
    mem = np.nonzero(pm)[0]
    assert len(mem) == 23552 and np.abs(pm[mem] + 0.07438609374462003).max() < 1e-5
    dev.close()


def sys_path_examples():
    import os, sys
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "idealized_geometries")
    if p not in sys.path:
        sys.path.insert(0, p)
