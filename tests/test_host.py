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
