"""CPU tests of the oracle: quadrature exactness, MMS convergence (the reference's only verification,
tests/run_MMS_space.py, restated WITH rate assertions), structural invariants, golden fixtures."""
import itertools
import os
from math import factorial

import numpy as np
import pytest

import knpemi_oracle as ko
import mms
from quadrature import simplex_rule
from common import synthetic_state, small_3d, relerr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("dim", [1, 2, 3])
def test_quadrature_exact(dim):
    for deg in range(0, 9):
        b, w = simplex_rule(dim, deg)
        assert abs(w.sum() - 1) < 1e-13
        for al in itertools.product(range(deg + 1), repeat=dim + 1):
            if sum(al) > deg:
                continue
            ex = factorial(dim) * np.prod([factorial(a) for a in al]) / factorial(dim + sum(al))
            num = (w * np.prod(b ** np.array(al), axis=1)).sum()
            assert abs(ex - num) < 5e-15, (dim, deg, al)


def _mms_errors(p, rs, dt, nsteps):
    errs = []
    for r in rs:
        pb = mms.build_space_mms(r, p=p, dt=dt)
        for _ in range(nsteps):
            ko.solve_for_time_step(pb, direct=True)
        errs.append(mms.l2_errors(pb, deg=5 if p == 1 else 8))
    return errs


def test_mms_space_p1_rates():
    """run_MMS_space.py: dt 1e-10, 2 steps, expected L2 order p+1 = 2."""
    errs = _mms_errors(1, range(2, 6), 1e-10, 2)
    for key in ("a", "b", "c", "phi"):
        rate = np.log(errs[-2][key] / errs[-1][key]) / np.log(2)
        assert rate > 1.9, (key, rate, errs)
    assert errs[-1]["phi"] < 4e-3 and errs[-1]["a"] < 1e-3


def test_mms_space_p2_rates():
    errs = _mms_errors(2, range(2, 5), 1e-10, 2)
    for key in ("a", "b", "c", "phi"):
        rate = np.log(errs[-2][key] / errs[-1][key]) / np.log(2)
        assert rate > 2.85, (key, rate, errs)


def test_mms_knp_operator_consistency():
    """Large dt (mass term no longer dominates): the KNP forms must still converge to the steady exact
    solution; an inconsistent diffusion / drift / upwind / coupling term would leave an O(1) error."""
    errs = _mms_errors(1, (3, 4), 1e-3, 10)
    for key in ("a", "b", "c"):
        assert np.log(errs[0][key] / errs[1][key]) / np.log(2) > 1.6, (key, errs)


@pytest.fixture(scope="module")
def pb2d():
    from knpemidg.mesh import make_mesh_2D
    m, s, f = make_mesh_2D(0)
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    synthetic_state(pb)
    return pb


def test_emi_operator_structure(pb2d):
    A, b, B = ko.assemble_emi(pb2d)
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    assert np.abs(A @ np.ones(A.shape[0])).max() < 1e-10 * abs(A).max()      # constants are the nullspace
    ev = np.linalg.eigvalsh(A.toarray())
    assert ev[0] > -1e-9 * ev[-1] and ev[1] > 1e-12 * ev[-1]                  # PSD with a single zero mode
    assert np.linalg.eigvalsh(B.toarray())[0] > 0                            # mass shift makes B definite


def test_knp_blocks_are_independent_and_nonsymmetric(pb2d):
    A0 = ko.assemble_knp(pb2d, 0)
    assert abs(A0 - A0.T).max() > 1e-6 * abs(A0).max()                       # drift + upwind
    # no membrane facet terms in a_knp: rows of cells that only touch the membrane couple to tag-0 neighbours only
    fc = pb2d.mesh.facet_cells[pb2d.mem]
    nd = pb2d.nd
    Ad = A0.tocsr()
    for c0, c1 in fc[:5]:
        blk = Ad[c0 * nd:(c0 + 1) * nd, c1 * nd:(c1 + 1) * nd]
        assert abs(blk).max() == 0


def test_rest_state_stays_at_rest():
    """Calibrated ICs of run_3D.py:80-86 with no stimulus: phi_M = -74.386 mV persists, c unchanged,
    electroneutrality holds (SURVEY.md section 4 invariant 3)."""
    m, s, f = small_3d()
    pb = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
    c0 = pb.c.copy()
    E = ko.solve_for_time_step(pb, direct=True)
    assert relerr(pb.c, c0) < 1e-9
    assert np.abs(pb.phi_M[pb.mem] + 0.07438609374462003).max() < 1e-9
    z = [i["z"] for i in pb.ions]
    charge = z[0] * pb.c[0] + z[1] * pb.c[1] + z[2] * pb.c_elim
    assert np.abs(charge).max() < 1e-9
    P = ko.idealized_params()
    EK = P["R"] * P["temperature"] / P["F"] * np.log(P["init"]["K"][1] / P["init"]["K"][0])
    assert np.abs(E["K"] - EK).max() < 1e-12 and abs(EK + 0.0936) < 2e-4       # SURVEY: E_K ~ -93.6 mV


@pytest.mark.parametrize("name", ["idealized_2D_r0", "box_3D_8x4x4", "box_3D_6x3x3_P2", "idealized_2D_r0_P2"])
def test_oracle_matches_golden(name):
    from knpemidg.mesh import Mesh
    g = np.load(os.path.join(GOLD, name + ".npz"))
    mesh = Mesh(g["coords"], g["cells"])
    assert np.array_equal(mesh.facet_cells, g["facet_cells"]) and np.array_equal(mesh.facet_local, g["facet_local"])
    pb = ko.build_idealized(mesh, g["cell_tags"], g["facet_tags"], p=int(g["degree"]) if "degree" in g else 1, membrane_tags=(1,))
    pb.c, pb.c_prev_n, pb.c_elim, pb.phi, pb.phi_M = g["c"], g["c_prev"], g["c_elim"], g["phi"], g["phi_M"]
    for k, ion in enumerate(pb.ions):
        pb.I_ch[ion["name"]] = g["I_ch"][k]
    A, b, _ = ko.assemble_emi(pb, want_B=False)
    assert relerr(A @ g["x"][0].ravel(), g["emi_Ax"]) < 1e-13
    assert relerr(b, g["emi_rhs"]) < 1e-13
    for k in range(pb.N_ions):
        assert relerr(ko.assemble_knp(pb, k) @ g["x"][k].ravel(), g["knp_Ax"][k]) < 1e-13
        assert relerr(ko.knp_rhs(pb, k), g["knp_rhs"][k]) < 1e-13
    ko.solve_for_time_step(pb, direct=True)
    assert relerr(pb.phi - pb.phi.mean(), g["step_phi"]) < 1e-8
    assert relerr(pb.c, g["step_c"]) < 1e-10


def test_iterative_matches_direct(pb2d):
    import copy
    a = copy.deepcopy(pb2d)
    b = copy.deepcopy(pb2d)
    ko.solve_emi(a, direct=True)
    ko.solve_emi(b, direct=False, rtol=1e-10, x0=np.zeros(b.ndof))
    assert relerr(a.phi - a.phi.mean(), b.phi - b.phi.mean()) < 1e-6
    b.phi = a.phi.copy()
    ko.solve_knp(a, direct=True)
    ko.solve_knp(b, direct=False, rtol=1e-12)
    assert relerr(a.c, b.c) < 1e-8


@pytest.mark.parametrize("which", ["2D", "3D"])
def test_oracle_connectivity_agrees_with_the_mesh_tables(which):
    """oracle/connectivity.py (dictionary matching, no facet tables) against the facet tables the oracle's forms run on
    (Mesh.facet_cells / facet_local, facet_orientation): two derivations of the same integer tables, equal entry for entry."""
    import connectivity as oc
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D
    mesh, sub, surf = make_mesh_2D(0) if which == "2D" else make_mesh_3D(0, n_axons=4)
    mtags = (1,) if which == "2D" else (1, 2, 3, 4)
    T = oc.derive_tables(mesh.cells, sub.array(), mesh.facets, surf.array(), mtags)
    fc, fl = mesh.facet_cells.astype(np.int64), mesh.facet_local.astype(np.int64)
    e_side = ko.facet_orientation(mesh, sub.array())
    for f in np.nonzero(fc[:, 1] >= 0)[0]:
        (c0, c1), (l0, l1) = fc[f], fl[f]
        assert T["nbr"][c0, l0] == c1 and T["nbr"][c1, l1] == c0 and T["nloc"][c0, l0] == l1 and T["nloc"][c1, l1] == l0
        assert T["fid"][c0, l0] == f and T["fid"][c1, l1] == f
        assert T["plus"][c0, l0] == (e_side[f] == 0) and T["plus"][c1, l1] == (e_side[f] == 1)
    assert (T["nbr"] < 0).sum() == (fc[:, 1] < 0).sum()
    assert len(T["mem"]) == np.isin(surf.array()[fc[:, 1] >= 0], mtags).sum() > 0
