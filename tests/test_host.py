"""CPU tests of the host side: mesh tables, membrane model / ODE integrator, the C-ABI library loads and
exports every symbol include/knpemi_hip.h declares (no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mesh_counts_match_reference_recipes():
    """Sizes derived from make_mesh_3D.py:81-105 / make_mesh_2D.py:75-86."""
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D, make_mesh_MMS
    m, s, f = make_mesh_3D(0)
    assert m.num_cells() == 15552 and len(m.interior_facets()) == 29790
    assert ((f.array() == 1) | (f.array() == 2)).sum() == 1472
    fc = m.facet_cells[(f.array() == 1) | (f.array() == 2)]
    assert (fc[:, 1] >= 0).all() and (s.array()[fc[:, 0]] != s.array()[fc[:, 1]]).all()
    m, s, f = make_mesh_2D(2)
    assert m.num_cells() == 3968 and len(m.interior_facets()) == 5820 and (f.array() == 1).sum() == 248
    m, s, f = make_mesh_MMS(3)
    assert m.num_cells() == 128 and [(f.array() == k).sum() for k in (1, 2, 3, 4)] == [4, 4, 4, 4]


def test_facet_table_consistency():
    from knpemidg.mesh import make_mesh_3D
    m, _, _ = make_mesh_3D(0, n_axons=1)
    assert (np.diff(m.cells, axis=1) > 0).all()                      # ascending vertex ids
    for side in (0, 1):
        sel = m.facet_cells[:, side] >= 0
        c = m.facet_cells[sel, side]
        l = m.facet_local[sel, side].astype(int)
        assert np.array_equal(m.cell_facets[c, l], np.nonzero(sel)[0])
        # facet vertices == cell vertices with local vertex l removed, in order
        cv = m.cells[c]
        mask = np.ones_like(cv, dtype=bool)
        mask[np.arange(len(c)), l] = False
        assert np.array_equal(cv[mask].reshape(len(c), -1), m.facets[sel])
    assert (m.facet_cells[:, 0] < np.where(m.facet_cells[:, 1] < 0, 1 << 30, m.facet_cells[:, 1])).all()


def test_interface_normal_points_from_low_to_high_tag():
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.utils import interface_normal
    m, s, f = make_mesh_2D(1)      # r=0 is too coarse to resolve the ICS box (equal-tag membrane facets)
    ng = interface_normal(s, m)
    mem = np.nonzero(f.array() == 1)[0]
    fc = m.facet_cells[mem]
    e_cell = fc[np.arange(len(mem)), ng.plus_side[mem]]
    assert (s.array()[e_cell] == 0).all()
    cm = m.cell_midpoints()
    i_cell = fc[np.arange(len(mem)), 1 - ng.plus_side[mem]]
    d = cm[i_cell] - cm[e_cell]
    assert (np.einsum("fd,fd->f", ng.vector[mem], d) > 0).all()


def test_constant_and_meshfunction():
    from knpemidg.mesh import Constant, MeshFunction, make_mesh_2D
    t = Constant(0.0)
    t.assign(float(t + 1e-4))
    assert abs(float(t) - 1e-4) < 1e-18
    m, s, f = make_mesh_2D(0)
    assert len(s.where_equal(1)) == (s.array() == 1).sum() and f.dim() == 1 and s.dim() == 2


def test_hh_integrator_matches_scipy():
    """Batched Dormand-Prince vs scipy LSODA (the reference integrates with LSODA rtol 1e-8, membrane.py:108-112)."""
    from scipy.integrate import solve_ivp
    from knpemidg.models import mm_hh
    from knpemidg.membrane import integrate_batch
    n = 5
    st = np.array([mm_hh.init_state_values() for _ in range(n)])
    pr = np.array([mm_hh.init_parameter_values() for _ in range(n)])
    pr[:, mm_hh.parameter_indices("Cm")] = 0.02
    pr[:, mm_hh.parameter_indices("E_Na")] = 0.0533
    pr[:, mm_hh.parameter_indices("E_K")] = -0.0936
    pr[:, mm_hh.parameter_indices("K_e")] = 3.32
    pr[:, mm_hh.parameter_indices("Na_i")] = 12.8
    pr[:, mm_hh.parameter_indices("stim_amplitude")] = np.linspace(0, 40, n)
    y = st.copy()
    p = pr.copy()
    for k in range(20):
        y, _ = integrate_batch(mm_hh.rhs, k * 1e-4, (k + 1) * 1e-4, y, p)
    for row in range(n):
        prow = pr[row:row + 1].copy()
        sol = solve_ivp(lambda t, s: mm_hh.rhs(t, s[None, :], prow)[0], (0, 20e-4), st[row], method="LSODA",
                        rtol=1e-10, atol=1e-13)
        assert np.abs(sol.y[:, -1] - y[row]).max() < 1e-6 * max(1.0, np.abs(y[row]).max())
    assert y[-1, 3] > y[0, 3] + 1e-3          # stimulus depolarises


def test_membrane_model_protocol():
    from knpemidg.mesh import make_mesh_2D
    from knpemidg.functions import FacetSpace, FacetFunction
    from knpemidg.membrane import MembraneModel
    from knpemidg.models import mm_hh
    m, s, f = make_mesh_2D(0)
    Q = FacetSpace(m)
    mm = MembraneModel(mm_hh, facet_f=f, tag=1, V=Q)
    assert mm.nodes == (f.array() == 1).sum()
    u = FacetFunction(Q)
    mm.get_membrane_potential(u)
    assert np.allclose(u.array()[mm.indices], -0.07438609374462003) and (np.delete(u.array(), mm.indices) == 0).all()
    mm.set_parameter_values({'Cm': lambda x: 0.02})
    g = FacetFunction(Q, np.full(Q.dim(), 5.0))
    mm.set_parameter('K_e', g)
    assert (mm.parameters[:, mm_hh.parameter_indices('K_e')] == 5.0).all()
    mm.set_parameter('Na_i', FacetFunction(Q, np.full(Q.dim(), 12.0)))
    mm.set_parameter('E_K', FacetFunction(Q, np.full(Q.dim(), -0.09)))
    mm.set_parameter('E_Na', FacetFunction(Q, np.full(Q.dim(), 0.05)))
    mm.step_lsoda(dt=1e-4, stimulus={'stim_amplitude': 10}, stimulus_locator=lambda x: x[0] < 20e-6)
    stim = mm.dof_locations[:, 0] < 20e-6
    assert (mm.parameters[stim, mm_hh.parameter_indices('stim_amplitude')] == 10).all()
    assert (mm.parameters[~stim, mm_hh.parameter_indices('stim_amplitude')] == 0).all()
    assert abs(mm.time - 1e-4) < 1e-18


def test_abi_library_exports_every_declared_symbol():
    import build as _b
    _b.build()
    from knpemidg import _abi
    lib = _abi.load()
    hdr = open(os.path.join(ROOT, "include", "knpemi_hip.h")).read()
    declared = set(re.findall(r"\b(knp_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("knp_ctx")
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "symbol %s declared in include/knpemi_hip.h is not exported" % name
        assert name in _abi.SIGNATURES, "no ctypes signature for %s" % name
    assert set(_abi.SIGNATURES) == declared


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under knp-emi-dg_amd/ may import or execute it."""
    pkg = os.path.join(ROOT, "knp-emi-dg_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "knpemi_oracle" not in txt and "import mms" not in txt and "oracle/" not in txt.replace("oracle/quadrature.py", ""), fn


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import build as _b
    _b.build()
    from knpemidg import _abi
    from knpemidg.mesh import make_mesh_2D
    m, s, f = make_mesh_2D(0)
    with pytest.raises(_abi.KnpError):
        _abi.Device(m, s.array(), f.array(), [1], 3)


def test_h5lite_reads_the_emix_mesh():
    """The pure-Python HDF5 subset reader against the reference's bundled EMIx mesh (chunked + deflate datasets; sizes from its
    XDMF descriptor: 22 419 vertices, 121 617 tets, labels 1..6)."""
    import os
    from knpemidg.h5lite import read_xdmf_mesh, H5File
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "examples", "emix_simulations", "meshes", "volume_ncells_5_size_5000")
    coords, cells, attrs = read_xdmf_mesh(os.path.join(d, "mesh.xdmf"))
    assert coords.shape == (22419, 3) and coords.dtype == np.float64
    assert cells.shape == (121617, 4) and cells.min() == 0 and cells.max() == 22418
    lab = attrs["label"]
    assert lab.shape == (121617,) and sorted(np.unique(lab)) == [1, 2, 3, 4, 5, 6]
    x = coords[cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
    assert vol.min() > 0 and 0.9 < vol.sum() / np.prod(coords.max(0) - coords.min(0)) <= 1.0
    assert set(H5File(os.path.join(d, "mesh.h5")).datasets) == {"data0", "data1", "data2"}
    with pytest.raises(KeyError):
        H5File(os.path.join(d, "mesh.h5")).read("nope")


def test_quadrature_rules_are_exact():
    """Product quadrature (feeds the device tabulations): every rule integrates all monomials up to its degree exactly
    (Dirichlet formula int lambda^alpha = d! alpha! / (|alpha| + d)!)."""
    import itertools
    from math import factorial
    from knpemidg.quadrature import simplex_rule
    for dim in (1, 2, 3):
        for deg in range(1, 11):
            bary, w = simplex_rule(dim, deg)
            assert abs(w.sum() - 1.0) < 1e-13 and (bary.sum(axis=1) - 1 < 1e-13).all()
            for alpha in itertools.product(range(deg + 1), repeat=dim + 1):
                if sum(alpha) > deg:
                    continue
                exact = factorial(dim) * np.prod([factorial(a) for a in alpha]) / factorial(sum(alpha) + dim)
                assert abs((w * np.prod(bary ** np.array(alpha), axis=1)).sum() - exact) < 1e-13, (dim, deg, alpha)


def test_dg_tabulations():
    """Tables uploaded by knp_set_tabulation: partition of unity, nodal property, facet tabulations live on the facet."""
    from knpemidg import dgtab
    for dim in (2, 3):
        nd = (dim + 1) * (dim + 2) // 2
        tabs = dgtab.tables(dim, 2)
        assert sorted(tabs) == list(range(11))
        for slot, (nloc, nq, w, B, dB) in tabs.items():
            assert B.shape == (nloc, nq, nd) and dB.shape == (nloc, nq, nd, dim + 1) and abs(w.sum() - 1) < 1e-13
            # sum_a phi_a = 1 on the simplex; its barycentric derivative is the same for every l (grad lambda_l sum to zero)
            assert np.abs(B.sum(axis=2) - 1).max() < 1e-13 and np.ptp(dB.sum(axis=2), axis=-1).max() < 1e-12
            if nloc > 1:
                for i in range(nloc):           # vertex i's basis function and every edge function touching i vanish on facet i
                    assert np.abs(B[i, :, i]).max() < 1e-14
        nodes = np.concatenate([np.eye(dim + 1)] + [[0.5 * (np.eye(dim + 1)[a] + np.eye(dim + 1)[b])] for a, b in dgtab.edges(dim + 1)])
        assert np.abs(dgtab.tabulate(2, nodes)[0] - np.eye(nd)).max() < 1e-14


def test_conforming_p2_auxiliary_space_is_galerkin():
    """The conforming-P2 operator assembled by knpemidg.amg equals P^T A P of the oracle's DG-P2 matrix (EMI with membrane
    coupling; KNP mass + diffusion), and its P1 coarse level is the Galerkin product through the P1->P2 interpolation."""
    import scipy.sparse as sp
    import knpemi_oracle as ko
    from knpemidg import amg
    from common import small_3d, synthetic_state
    m, s, f = small_3d((6, 3, 3))
    pb = ko.build_idealized(m, s.array(), f.array(), p=2, membrane_tags=(1,))
    synthetic_state(pb)
    cs = amg.ConformingSpace(m, f.array(), (1,))
    c2 = amg.ConformingSpaceP2(cs)
    P = sp.coo_matrix((np.ones(pb.ndof), (np.arange(pb.ndof), c2.dof.ravel())), shape=(pb.ndof, c2.n)).tocsr()
    A, _, _ = ko.assemble_emi(pb, want_B=False)
    Ac = c2.stiffness(pb.kappa(), membrane=(pb.mem, pb.C_phi))
    ref = (P.T @ A @ P).tocsr()
    assert abs(Ac - ref).max() < 1e-12 * abs(ref).max()
    pb.phi[:] = 0.0
    Ak = ko.assemble_knp(pb, 0)
    Ack = c2.stiffness(pb.ions[0]["D"], mass_coef=np.full(m.num_cells(), 1.0 / pb.dt))
    refk = (P.T @ Ak @ P).tocsr()
    assert abs(Ack - refk).max() < 1e-12 * abs(refk).max()
    H = amg.build_hierarchy(Ac, top_interp=c2.interp, max_coarse=40)
    assert [lv.A.shape[0] for lv in H][:2] == [c2.n, cs.n]
    assert abs(H[1].A - c2.interp.T @ Ac @ c2.interp).max() < 1e-12 * abs(H[1].A).max()
    # P1 functions are reproduced: interpolating a linear field gives its values at the P2 nodes
    lin = m.coords @ np.array([1.0, -2.0, 0.5])
    v1 = np.zeros(cs.n); v1[cs.dof.ravel()] = lin[m.cells].ravel()
    x = m.coords[m.cells]
    mids = np.stack([0.5 * (x[:, a] + x[:, b]) for a, b in c2.edges], axis=1) @ np.array([1.0, -2.0, 0.5])
    v2 = c2.interp @ v1
    assert np.abs(v2[c2.dof[:, 4:]] - mids).max() < 1e-12 * np.abs(lin).max()


def test_h5_writer_round_trip_and_libhdf5(tmp_path):
    """Result-file writer (knpemidg.h5lite.H5Writer: nested groups, contiguous datasets) read back by the package's own reader
    and -- where a libhdf5 is installed -- by the real HDF5 library through ctypes."""
    import ctypes as C
    import glob
    from knpemidg.h5lite import H5Writer, H5File
    rng = np.random.default_rng(0)
    data = {"/mesh/coordinates": rng.standard_normal((1000, 3)), "/mesh/topology": rng.integers(0, 1000, size=(500, 4)).astype(np.int64),
            "/subdomains/values": rng.integers(0, 5, size=500).astype(np.uint64), "/f32": np.arange(5, dtype=np.float32),
            "/i32": np.arange(-3, 4, dtype=np.int32)}
    for n in range(40):
        data["/potential/vector_%d" % n] = rng.standard_normal(77)
    path = str(tmp_path / "t.h5")
    with H5Writer(path) as w:
        for k, v in data.items():
            w.write(k, v)
    f = H5File(path)
    assert sorted(f.datasets) == sorted(k.lstrip("/") for k in data)
    for k, v in data.items():
        got = f.read(k)
        assert got.dtype == v.dtype and np.array_equal(got, v)
    libs = sorted(glob.glob("/opt/conda/lib/libhdf5.so*") + glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    if not libs:
        pytest.skip("no libhdf5 on this machine for the cross-check")
    lib = C.CDLL(libs[0])
    lib.H5open()
    hid = C.c_int64
    lib.H5Fopen.restype = hid; lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Dopen2.restype = hid; lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dget_space.restype = hid; lib.H5Dget_space.argtypes = [hid]
    lib.H5Sget_simple_extent_ndims.argtypes = [hid]
    lib.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    fid = lib.H5Fopen(path.encode(), 0, 0)
    assert fid >= 0
    native = {np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g", np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.int64): "H5T_NATIVE_INT64_g",
              np.dtype(np.uint64): "H5T_NATIVE_UINT64_g", np.dtype(np.int32): "H5T_NATIVE_INT32_g"}
    for k in ("/mesh/coordinates", "/mesh/topology", "/subdomains/values", "/f32", "/i32", "/potential/vector_39"):
        d = lib.H5Dopen2(fid, k.encode(), 0)
        assert d >= 0, k
        sp = lib.H5Dget_space(d)
        nd = lib.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * nd)()
        lib.H5Sget_simple_extent_dims(sp, dims, None)
        a = np.empty(tuple(dims), dtype=data[k].dtype)
        assert lib.H5Dread(d, hid.in_dll(lib, native[data[k].dtype]).value, 0, 0, 0, a.ctypes.data) >= 0
        assert np.array_equal(a, data[k]), k


def test_dolfin_xml_mesh_round_trip(tmp_path):
    """DOLFIN XML mesh / MeshFunction files (run_3D.py:166-168) written and read back: coordinates, connectivity, cell tags and
    facet tags keyed by (cell, local facet) land on the same entities."""
    from knpemidg.mesh import make_mesh_3D, make_mesh_2D
    from knpemidg import mesh_io
    from common import small_3d
    for mt in (small_3d((6, 3, 3)), make_mesh_2D(0)):
        m, s, f = mt
        mesh_io.write_dolfin_xml_mesh(m, str(tmp_path / "mesh.xml"))
        mesh_io.write_dolfin_xml_meshfunction(m, s, str(tmp_path / "sub.xml"))
        mesh_io.write_dolfin_xml_meshfunction(m, f, str(tmp_path / "surf.xml"))
        m2 = mesh_io.read_dolfin_xml_mesh(str(tmp_path / "mesh.xml"))
        assert np.array_equal(m2.cells, m.cells) and np.allclose(m2.coords, m.coords, rtol=1e-15, atol=0)
        assert np.array_equal(m2.facet_cells, m.facet_cells)
        s2 = mesh_io.read_dolfin_xml_meshfunction(m2, str(tmp_path / "sub.xml"))
        f2 = mesh_io.read_dolfin_xml_meshfunction(m2, str(tmp_path / "surf.xml"))
        assert np.array_equal(s2.array(), s.array()) and np.array_equal(f2.array(), f.array())


@pytest.mark.parametrize("degree", [1, 2])
def test_knp_hierarchy_helper_process_matches_in_process(degree):
    """knpemidg/setup_worker.py: the helper process builds the same KNP hierarchies from the pickled job as the solver's process
    does from its own mesh (levels, operators and smoother data bit for bit), and a failing helper reports instead of hanging."""
    from common import small_3d
    from knpemidg import amg, setup_worker
    mesh, sub, surf = small_3d((8, 4, 4))
    D_subs = [{0: 1.33e-9, 1: 1.33e-9}, {0: 1.96e-9, 1: 1.96e-9}]
    cs = amg.ConformingSpace(mesh, surf.array(), [1])
    cs2 = amg.ConformingSpaceP2(cs) if degree == 2 else None
    ref = amg.build_knp_groups(cs, cs2, sub.array(), D_subs, 1e-4, 1)
    handle = setup_worker.start(setup_worker.job_from_solver(mesh, sub.array(), surf.array(), [1], degree, D_subs, 1e-4, 1))
    got = setup_worker.collect(handle)
    assert got is not None and len(got) == len(ref)
    for (m0, l0), (m1, l1) in zip(ref, got):
        assert m0 == m1 and len(l0) == len(l1)
        for a, b in zip(l0, l1):
            assert a.A.shape == b.A.shape and (a.A != b.A).nnz == 0
            for name in ("P", "R"):
                pa, pb = getattr(a, name, None), getattr(b, name, None)
                assert (pa is None) == (pb is None)
                if pa is not None:
                    assert (pa != pb).nnz == 0
    # the first EMI hierarchy the same way
    rng = np.random.default_rng(3)
    nd = 4 if degree == 1 else 10
    kappa = rng.uniform(0.5, 1.5, size=(mesh.num_cells(), nd))
    ref_e = amg.build_emi_levels(cs, cs2, surf.array(), [1], kappa, 2.0e2)
    res = setup_worker.collect(setup_worker.start(setup_worker.emi_job(mesh, surf.array(), [1], degree, kappa, 2.0e2)))
    assert res is not None and len(res["levels"]) == len(ref_e)
    assert np.array_equal(res["dof"], (cs2 if cs2 is not None else cs).dof)
    for a, b in zip(ref_e, res["levels"]):
        assert a.A.shape == b.A.shape and (a.A != b.A).nnz == 0
    bad = setup_worker.start({"coords": None})
    assert setup_worker.collect(bad) is None


def test_emi_dg_smoother_is_chosen_by_measurement():
    """knpemidg/solver.py: Solver._emi_dg_chebyshev / _emi_smoother_trial -- round 3 read the DG-level smoother of the EMI preconditioner
    off mesh-size thresholds; round 4 measures it on the first EMI system of the run: the same system is solved from the same initial
    guess with and without the Chebyshev step (one untimed solve each first), each charged its time per decade of true-residual
    reduction, the costs all-reduced (every rank of a partitioned run takes the same decision), the step dropped only if that is
    >= 3 % cheaper; the step's solution comes from a solve with the chosen variant; solver_params / KNP_EMI_CHEB decide explicitly."""
    from collections import namedtuple
    from knpemidg.solver import Solver

    class Dev:
        def __init__(self):
            self.calls, self.reduced, self.uploads = [], [], 0

        def set_emi_dg_smoother(self, on):
            self.calls.append(bool(on))

        def allreduce_sum(self, v):
            self.reduced.append(list(v))
            return np.asarray(v, dtype=float) * 3.0          # three ranks with the same timings

        def download(self, field):
            return np.zeros(4)

        def upload(self, field, a):
            self.uploads += 1

    def run(cost_on, cost_off, explicit=None):
        S = Solver.__new__(Solver)
        S.verbose = False
        S.dev = Dev()
        if explicit is not None:
            S.solver_params = namedtuple("solver_params", ("emi_dg_chebyshev",))(explicit)
        first = S._emi_dg_chebyshev()
        if S._emi_trial is None:
            return first, S.dev.calls, None, None

        def solve():                                         # one decade of reduction: seconds = cost per decade
            on = S.dev.calls[-1]
            return (cost_on if on else cost_off), (7 if on else 9), [1.0, 0.1, 0.0]
        out = S._emi_smoother_trial(solve)
        assert S._emi_trial is None and len(S.dev.reduced) == 1
        return first, S.dev.calls, S.emi_dg_chebyshev_measured, out

    first, calls, m, out = run(1.0, 0.8)                      # plain block-Jacobi 20 % cheaper per decade: dropped
    assert first is True and calls == [True, False, True, False] and m["chosen"] is False and out[1] == 9
    first, calls, m, out = run(1.0, 0.99)                     # within the 3 % margin: the step stays, the step's solve is redone with it
    assert calls == [True, False, True, False, True] and m["chosen"] is True and out[1] == 7
    first, calls, m, out = run(1.0, 1.25)
    assert calls[-1] is True and abs(m["plain_s_per_decade"] - 3.0 * 1.25) < 1e-12
    assert run(None, None, explicit=False)[0] is False and run(None, None, explicit=True)[0] is True      # explicit: no trial
    os.environ["KNP_EMI_CHEB"] = "0"
    try:
        assert run(None, None)[0] is False
    finally:
        del os.environ["KNP_EMI_CHEB"]


def test_native_setup_kernels_match_numpy():
    """Host setup kernels of the library (csrc/host_sparse.cpp, round 4) against the numpy code they replace: facet table (identical
    numbering), geometry classes (same grouping, same records), cell Gram matrices, the CSR pattern of the conforming operators
    (identical to a stable argsort of the entry keys) and the scatter-add into it."""
    import build as _b
    _b.build()
    from knpemidg import _abi, amg
    from knpemidg import mesh as M
    m, s, f = M.make_mesh_3D(1)                                  # 124 416 tets: above the size from which the native paths are taken
    ref = M.Mesh.__new__(M.Mesh)
    ref.coords, ref.cells, ref.gdim = m.coords, m.cells, 3
    native = M.Mesh._build_facets_native
    M.Mesh._build_facets_native = lambda self: False
    try:
        ref._build_facets()
    finally:
        M.Mesh._build_facets_native = native
    for a in ("cell_facets", "facets", "facet_cells", "facet_local"):
        assert np.array_equal(getattr(m, a), getattr(ref, a)) and getattr(m, a).dtype == getattr(ref, a).dtype, a
    order = _abi.morton_order(m.cell_midpoints())
    g_nat = _abi.geometry_classes(m, order)
    keep = _abi._geometry_classes_native
    _abi._geometry_classes_native = lambda *a: None
    try:
        g_np = _abi.geometry_classes(m, order)
    finally:
        _abi._geometry_classes_native = keep
    pairs = np.unique(np.stack([g_nat[0].astype(int), g_np[0].astype(int)], axis=1), axis=0)
    assert len(pairs) == g_nat[1].shape[0] == g_np[1].shape[0] == 24          # same grouping: a bijection between the class ids
    perm = np.array([dict(pairs.tolist())[i] for i in range(24)])
    assert np.array_equal(g_nat[1], g_np[1][perm])
    cs = amg.ConformingSpace(m, f.array(), (1, 2))
    vol, G = amg._cell_gram(cs)
    x = m.coords[m.cells]
    J = (x[:, 1:, :] - x[:, :1, :]).transpose(0, 2, 1)
    Ji = np.linalg.inv(J)
    g = np.empty((x.shape[0], 4, 3))
    g[:, 1:, :] = Ji
    g[:, 0, :] = -Ji.sum(axis=1)
    assert np.abs(vol - np.abs(np.linalg.det(J)) / 6.0).max() < 1e-14 * vol.max()
    assert np.abs(G - np.einsum("cad,cbd->cab", g, g)).max() < 1e-13 * np.abs(G).max()
    kappa = np.random.default_rng(0).uniform(0.5, 1.5, size=(m.num_cells(), 4))
    A = cs.stiffness(kappa)
    dof = cs.dof.astype(np.int64)
    key = (np.repeat(dof[:, :, None], 4, axis=2) * cs.n + np.repeat(dof[:, None, :], 4, axis=1)).ravel()
    o = np.argsort(key, kind="stable")
    ks = key[o]
    starts = np.nonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))[0]
    pat = cs._pattern
    assert np.array_equal(pat[0], o) and np.array_equal(pat[1], starts) and np.array_equal(pat[2], (ks[starts] % cs.n).astype(np.int32))
    blk = cs.cell_blocks(kappa)
    want = np.add.reduceat(blk.ravel()[o], starts)
    assert np.abs(A.data - want).max() < 1e-15 * np.abs(want).max()             # the same sums up to their order


def test_native_mesh_and_hierarchy_kernels_match_numpy(monkeypatch):
    """More host kernels of the library (csrc/host_sparse.cpp) against the numpy passes they replace, on the 124 416-tet mesh: the Morton
    orders of cells and vertices and the median cell extent (identical), the box tags of the mesh generator (identical), the neighbour
    table behind the geometry classes (identical), the distance-2 MIS aggregation (identical aggregates) and the mirrored fp32 coarse
    inverse (identical to tril + transpose + astype)."""
    import build as _b
    _b.build()
    from knpemidg import _abi, amg
    from knpemidg import mesh as M
    m, s, f = M.make_mesh_3D(1)
    nc = m.num_cells()
    lib = _abi.load()
    # Morton orders
    xc = m.coords[m.cells]
    scale = np.maximum(np.median(xc.max(axis=1) - xc.min(axis=1), axis=0), 1e-300)
    sc = np.empty(3)
    co, cl = np.ascontiguousarray(m.coords), np.ascontiguousarray(m.cells, dtype=np.int32)
    assert lib.knp_host_cell_extent_median(nc, 4, 3, _abi._p(co, _abi._f64p), _abi._p(cl, _abi._i32p), _abi._p(sc, _abi._f64p)) == 0
    assert np.array_equal(sc, scale)
    monkeypatch.setenv("KNP_SETUP_NATIVE_MORTON", "0")
    o_np, v_np = _abi.morton_order(m.cell_midpoints(), scale), _abi.morton_order(m.coords, scale)
    monkeypatch.setenv("KNP_SETUP_NATIVE_MORTON", "1")
    assert np.array_equal(_abi._morton_native(co, cl, scale), o_np)
    assert np.array_equal(_abi.morton_order(m.cell_midpoints(), scale), o_np)      # through the public function (points given)
    assert np.array_equal(_abi._morton_native(co, None, scale), v_np)
    # box tags of the generator
    keep = M._box_marks_native
    M._box_marks_native = lambda *a: (None, None)
    try:
        m0, s0, f0 = M.make_mesh_3D(1)
    finally:
        M._box_marks_native = keep
    assert np.array_equal(s0.array(), s.array()) and np.array_equal(f0.array(), f.array()) and np.array_equal(m0.cells, m.cells)
    # neighbour table
    fc, fl, cf = m.facet_cells, m.facet_local.astype(np.int64), m.cell_facets
    side = (fc[cf, 0] != np.arange(nc)[:, None]).astype(np.int64)
    nb = np.take_along_axis(fc[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]
    nj = np.take_along_axis(fl[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]
    nb2, nj2 = np.empty((nc, 4), np.int32), np.empty((nc, 4), np.int8)
    assert lib.knp_host_cell_neighbours(nc, 4, _abi._p(np.ascontiguousarray(cf, dtype=np.int32), _abi._i32p),
                                        _abi._p(np.ascontiguousarray(fc, dtype=np.int32), _abi._i32p),
                                        _abi._p(np.ascontiguousarray(m.facet_local, dtype=np.int8), _abi._i8p), _abi._p(nb2, _abi._i32p),
                                        _abi._p(nj2, _abi._i8p), 0) == 0
    assert np.array_equal(nb, nb2) and np.array_equal(nj, nj2)
    # MIS-2 aggregation on the strength graph of the conforming operator
    cs = amg.ConformingSpace(m, f.array(), (1, 2))
    A = cs.stiffness(np.ones((nc, 4))).tocsr()
    A.setdiag(0.0)
    A.eliminate_zeros()
    S = (abs(A) > 0).astype(np.float64).tocsr()
    S.sort_indices()
    assert S.shape[0] >= 20000
    a1, n1 = amg.mis2_aggregate(S, seed=1)
    monkeypatch.setenv("KNP_SETUP_NATIVE_MIS2", "0")
    a0, n0 = amg.mis2_aggregate(S, seed=1)
    assert n0 == n1 and np.array_equal(a0, a1) and a1.min() == 0 and len(np.unique(a1)) == n1
    # mirrored + rounded coarse inverse
    rng = np.random.default_rng(2)
    N = 300
    Mf = np.asfortranarray(rng.standard_normal((N, N)))
    L = np.tril(Mf)
    want = (L + L.T)
    want[np.diag_indices(N)] *= 0.5
    got, mx = amg._mirror_round(Mf)
    assert got.dtype == np.float32 and np.array_equal(got, want.astype(np.float32)) and mx == np.abs(want).max()
    Mf[7, 3] = np.nan
    assert not (amg._mirror_round(Mf)[1] <= 1e300)
    # the deflated coarse inverse itself: pseudo-inverse of a singular graph Laplacian
    import scipy.sparse as sp
    n = 400
    i = np.arange(n - 1)
    W = sp.coo_matrix((rng.uniform(0.5, 2.0, n - 1), (i, i + 1)), shape=(n, n))
    W = (W + W.T).tocsr()
    Lap = (sp.diags(np.asarray(W.sum(axis=1)).ravel()) - W).tocsr()
    P = amg._coarse_pseudo_inverse(Lap, np.ones(n))
    ref = np.linalg.pinv(Lap.toarray(), hermitian=True)
    assert P.dtype == np.float32 and np.array_equal(P, P.T) and np.abs(P - ref).max() < 1e-6 * np.abs(ref).max()


def test_native_mesh_kernels_on_a_large_2d_mesh(monkeypatch):
    """The same host kernels in 2D (63 488 triangles: above the size from which the native paths are taken): facet table, box tags and
    Morton order identical to the numpy code."""
    import build as _b
    _b.build()
    from knpemidg import _abi
    from knpemidg import mesh as M
    m1 = M.make_mesh_2D(4)
    keep, nat = M._box_marks_native, M.Mesh._build_facets_native
    M._box_marks_native = lambda *a: (None, None)
    M.Mesh._build_facets_native = lambda self: False
    try:
        m0 = M.make_mesh_2D(4)
    finally:
        M._box_marks_native, M.Mesh._build_facets_native = keep, nat
    assert m1[0].num_cells() == 63488
    for a in ("cells", "facets", "facet_cells", "facet_local", "cell_facets"):
        assert np.array_equal(getattr(m0[0], a), getattr(m1[0], a)), a
    assert np.array_equal(m0[1].array(), m1[1].array()) and np.array_equal(m0[2].array(), m1[2].array())
    mesh = m1[0]
    xc = mesh.coords[mesh.cells]
    scale = np.maximum(np.median(xc.max(axis=1) - xc.min(axis=1), axis=0), 1e-300)
    monkeypatch.setenv("KNP_SETUP_NATIVE_MORTON", "0")
    o0 = _abi.morton_order(mesh.cell_midpoints(), scale)
    monkeypatch.setenv("KNP_SETUP_NATIVE_MORTON", "1")
    co, cl = np.ascontiguousarray(mesh.coords), np.ascontiguousarray(mesh.cells, dtype=np.int32)
    sc = np.empty(2)
    assert _abi.load().knp_host_cell_extent_median(mesh.num_cells(), 3, 2, _abi._p(co, _abi._f64p), _abi._p(cl, _abi._i32p), _abi._p(sc, _abi._f64p)) == 0
    assert np.array_equal(sc, scale) and np.array_equal(_abi._morton_native(co, cl, scale), o0)


def test_helper_transport_out_of_band_and_in_band(monkeypatch):
    """knpemidg/setup_worker.py: jobs and results travel as a small pickle + one file of array bytes under /dev/shm (pickle protocol 5,
    out-of-band buffers, mapped copy-on-write by the receiver and unlinked at once); in band when /dev/shm is not there, when switched off,
    or for small payloads.  Same objects either way, 64-byte aligned and writable arrays, nothing left behind."""
    import glob
    import io
    import scipy.sparse as sp
    from knpemidg import setup_worker as W
    rng = np.random.default_rng(0)
    A = sp.random(3000, 3000, density=0.01, random_state=1, format="csr")
    obj = {"a": rng.standard_normal((70001, 3)), "odd": np.arange(13, dtype=np.int8), "b": rng.integers(0, 9, 250001).astype(np.int32), "A": A,
           "f32": rng.standard_normal(100003).astype(np.float32), "text": "x", "nested": [np.arange(5), {"k": 1.5}]}

    def same(x, y):
        assert np.array_equal(x["a"], y["a"]) and np.array_equal(x["odd"], y["odd"]) and np.array_equal(x["b"], y["b"])
        assert abs(x["A"] - y["A"]).max() == 0 and np.array_equal(x["f32"], y["f32"]) and y["f32"].dtype == np.float32
        assert y["text"] == "x" and np.array_equal(y["nested"][0], np.arange(5)) and y["nested"][1] == {"k": 1.5}
    before = set(glob.glob("/dev/shm/knp_setup_*"))
    buf = io.BytesIO()
    W._dump(obj, buf)
    small = buf.tell()
    buf.seek(0)
    out = W._load(buf)
    same(obj, out)
    if os.path.isdir("/dev/shm"):
        assert small < 100000                                     # the arrays did not go through the stream
        for k in ("a", "b", "f32"):
            assert out[k].ctypes.data % 64 == 0 and out[k].flags.writeable
        out["a"][0, 0] = 123.0                                     # private mapping: edits stay local
    assert set(glob.glob("/dev/shm/knp_setup_*")) == before      # the receiver unlinked the file
    monkeypatch.setenv("KNP_SETUP_NO_SHM", "1")
    buf = io.BytesIO()
    W._dump(obj, buf)
    assert buf.tell() > 1000000
    buf.seek(0)
    same(obj, W._load(buf))
    monkeypatch.delenv("KNP_SETUP_NO_SHM")
    buf = io.BytesIO()
    W._dump({"tiny": np.arange(10)}, buf)                        # below 1 MB: in band
    buf.seek(0)
    assert np.array_equal(W._load(buf)["tiny"], np.arange(10))
    # a file nobody picked up is removed by the writer's cleanup
    W._dump(obj, io.BytesIO())
    W._shm_cleanup()
    assert set(glob.glob("/dev/shm/knp_setup_*")) == before


def test_native_smoothed_aggregation_passes_give_the_same_hierarchy(monkeypatch):
    """knp_host_strength / knp_host_smooth_prolongator / knp_host_truncate_prolongator (csrc/host_sparse.cpp) against the numpy / scipy lines
    of amg.build_hierarchy they replace: the EMI and KNP hierarchies of the 124 416-tet mesh come out bit for bit the same (matrices,
    prolongators, coarse inverse)."""
    import build as _b
    _b.build()
    from knpemidg import amg
    from knpemidg import mesh as M
    m, s, f = M.make_mesh_3D(1)
    nc = m.num_cells()
    kappa = np.random.default_rng(0).uniform(0.5, 1.5, (nc, 4))
    D = [{0: 1.33e-9, 1: 1.33e-9, 2: 1.33e-9}, {0: 2.03e-9, 1: 2.03e-9, 2: 2.03e-9}]          # 47 % apart: one hierarchy per species

    def build():
        cs = amg.ConformingSpace(m, f.array(), (1, 2))
        return amg.build_emi_levels(cs, None, f.array(), (1, 2), kappa, 1.0), amg.build_knp_groups(cs, None, s.array(), D, 1.0e-4, 1)
    monkeypatch.setenv("KNP_SETUP_NATIVE_SA", "1")
    emi1, knp1 = build()
    monkeypatch.setenv("KNP_SETUP_NATIVE_SA", "0")
    emi0, knp0 = build()
    assert emi1[0].A.shape[0] >= 20000 and len(emi1) >= 2 and len(knp1) == 2

    def same(la, lb):
        assert len(la) == len(lb)
        for x, y in zip(la, lb):
            for name in ("A", "P", "R"):
                if hasattr(x, name):
                    X, Y = getattr(x, name), getattr(y, name)
                    assert np.array_equal(X.indptr, Y.indptr) and np.array_equal(X.indices, Y.indices) and np.array_equal(X.data, Y.data), name
            assert np.array_equal(x.dinv, y.dinv) and x.rho == y.rho
        assert np.array_equal(la[-1].pinv, lb[-1].pinv)
    same(emi1, emi0)
    for (ma, la), (mb, lb) in zip(knp1, knp0):
        assert ma == mb
        same(la, lb)


def test_parallel_facet_builder_equals_the_serial_one(monkeypatch):
    """knp_host_build_facets above 2^21 (cell, local facet) pairs: the compare-and-swap build on several threads numbers the facets exactly
    like the serial loop (first appearance in (cell, local facet) order, side 0 = the first cell), in 3D and 2D, and still refuses a facet
    with three cells."""
    import build as _b
    _b.build()
    from knpemidg import _abi
    from knpemidg import mesh as M
    lib = _abi.load()

    def run(cells, threads):
        monkeypatch.setenv("KNP_SETUP_THREADS", str(threads))
        nc, nv = cells.shape
        cf, fa = np.empty((nc, nv), np.int32), np.empty((nc * nv, nv - 1), np.int32)
        fc, fl = np.empty((nc * nv, 2), np.int32), np.empty((nc * nv, 2), np.int8)
        nf = int(lib.knp_host_build_facets(nc, nv, _abi._p(cells, _abi._i32p), _abi._p(cf, _abi._i32p), _abi._p(fa, _abi._i32p), _abi._p(fc, _abi._i32p),
                                           _abi._p(fl, _abi._i8p)))
        return nf, cf, fa[:max(nf, 0)].copy(), fc[:max(nf, 0)].copy(), fl[:max(nf, 0)].copy()
    m3 = M.BoxMesh((0, 0, 0), (1, 1, 1), 48, 48, 40)                                                                      # 552 960 tets: 2.2 M pairs
    c3 = np.ascontiguousarray(m3.cells, dtype=np.int32)
    assert c3.shape[0] * 4 >= 1 << 21
    a, b = run(c3, 1), run(c3, 6)
    assert a[0] == b[0] > 0 and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))
    f = np.nonzero(m3.facet_cells[:, 1] >= 0)[0][4321]
    extra = np.sort(np.concatenate([m3.facets[f], [m3.coords.shape[0] - 1]])).astype(np.int32)
    bad = np.ascontiguousarray(np.vstack([c3, extra[None, :]]))
    assert run(bad, 1)[0] == -2 and run(bad, 6)[0] == -2
    m2 = M.RectangleMesh((0, 0), (4, 1), 700, 260, "crossed")                                                             # 728 000 triangles
    c2 = np.ascontiguousarray(m2.cells, dtype=np.int32)
    assert c2.shape[0] * 3 >= 1 << 21
    a, b = run(c2, 1), run(c2, 6)
    assert a[0] == b[0] > 0 and all(np.array_equal(x, y) for x, y in zip(a[1:], b[1:]))


def test_cancelled_helpers_leave_nothing_in_dev_shm():
    """setup_worker.cancel at any moment of a hand-over (before the helper mapped the job file, while it works, after it wrote its result file):
    both files are named by the parent and removed by it on every path."""
    import glob
    import time
    from knpemidg import setup_worker as W
    from knpemidg import mesh as M
    if not os.path.isdir("/dev/shm"):
        pytest.skip("no /dev/shm")
    m, s, f = M.make_mesh_3D(0)
    kappa = np.random.default_rng(0).uniform(0.5, 1.5, (m.num_cells(), 4))
    job = W.emi_job(m, f.array(), [1, 2], 1, kappa, 1.0)                    # 1.7 MB of arrays: out of band
    before = set(glob.glob("/dev/shm/knp_setup_*"))
    for delay in (0.0, 0.05, 0.4, 1.5):
        h = W.start(job)
        time.sleep(delay)
        W.cancel(h)
        time.sleep(0.2)
        assert set(glob.glob("/dev/shm/knp_setup_*")) == before, delay
    h = W.start(job)
    res = W.collect(h)
    assert res is not None and len(res["levels"]) >= 1
    assert set(glob.glob("/dev/shm/knp_setup_*")) == before
