"""Bit-exact parity of the connectivity / orientation tables (`north_star`: "bit-exact for DoF/connectivity indexing").

The library derives its neighbour, flag (neighbour-local facet, integration class, plus-side bit), facet-id, membrane-facet and
halo-block tables inside `knp_ctx_create` (csrc/abi.hip) from raw cells / tags -- the device counterpart of DOLFIN's facet
topology, of the `dS(tag)` classification (reference: src/knpemidg/solver.py:113-121) and of `interface_normal` / `plus` /
`minus` (reference: src/knpemidg/utils.py:61-98).  They are read back through `knp_debug_table` and compared ENTRY FOR ENTRY with
the tables `oracle/connectivity.py` derives independently (dictionary matching of sorted vertex tuples in the caller's numbering,
none of the product's facet tables).  Integer data: every comparison is `array_equal`."""
import os
import sys

import numpy as np
import pytest

import connectivity as oc
from knpemidg import _abi as A

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mesh(which):
    from knpemidg.mesh import make_mesh_2D, make_mesh_3D
    if which == "2D_r0":
        return make_mesh_2D(0) + ((1,),)
    if which == "3D_4axon_r0":
        return make_mesh_3D(0, n_axons=4) + ((1, 2, 3, 4),)
    sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
    from emix_common import load_mesh
    return load_mesh() + ((1, 2),)


def _expected_halo_tables(nbr_d, nloc_d, kind_d, nc_owned, B=256):
    """Layout contract of KNP_DT_HB_SRC / KNP_DT_HB_LOC (include/knpemi_hip.h), restated: per block of B consecutive device cells
    the coupled (SIPG or membrane) neighbours outside the block in (cell, facet) order."""
    nblk = (nc_owned + B - 1) // B
    loc = np.zeros((nc_owned, 4), dtype=np.uint16)
    lists = [[] for _ in range(nblk)]
    for k in range(nc_owned):
        b = k // B
        for a in range(4):
            n = nbr_d[k, a]
            if kind_d[k, a] not in (oc.K_SIPG, oc.K_MEMBRANE) or n < 0:
                continue
            if n // B == b:
                loc[k, a] = n - b * B
            else:
                loc[k, a] = B + len(lists[b])
                lists[b].append(4 * n + nloc_d[k, a])
    return loc, lists


@pytest.mark.parametrize("which", ["2D_r0", "3D_4axon_r0", "emix"])
def test_device_connectivity_tables_bit_exact(hip_lib, which):
    mesh, sub, surf, mtags = _mesh(which)
    ctags, ftags = sub.array(), surf.array()
    exp = oc.derive_tables(mesh.cells, ctags, mesh.facets, ftags, mtags)
    dev = A.Device(mesh, ctags, ftags, mtags, 3)
    nc, nv = mesh.cells.shape
    meta = dev.debug_table(A.DT_META)
    assert meta[0] == nc and meta[1] == nc and meta[2] == mesh.num_facets() and meta[7] == mesh.gdim
    order, rank = dev.cell_order, dev.cell_rank                      # device -> caller, caller -> device
    assert np.array_equal(np.sort(order), np.arange(nc)) and np.array_equal(rank[order], np.arange(nc))

    # neighbours: device ids -> caller ids, rows -> caller rows
    nbr_d = dev.debug_table(A.DT_NBR).reshape(nc, nv).astype(np.int64)
    nbr_c = np.where(nbr_d >= 0, order[np.maximum(nbr_d, 0)], -1)
    assert np.array_equal(nbr_c, exp["nbr"][order])

    # flag bytes
    flag = dev.debug_table(A.DT_FLAG)
    fb = ((flag[:, None] >> (8 * np.arange(nv, dtype=np.uint32))[None, :]) & 0xFF).astype(np.int64)
    e = {k: v[order] for k, v in exp.items() if k != "mem"}
    assert np.array_equal((fb >> 2) & 3, e["kind"])
    assert np.array_equal(fb & 3, e["nloc"])
    assert np.array_equal((fb >> 4) & 1, e["plus"])
    assert not (fb >> 5).any()
    if nv == 3:
        assert not (flag >> 24).any()                                 # the unused fourth byte of a triangle

    # facet ids (caller numbering)
    assert np.array_equal(dev.debug_table(A.DT_CFACET).reshape(nc, nv), e["fid"])

    # membrane facets, in facet order
    mf = dev.debug_table(A.DT_MF).reshape(-1, 6).astype(np.int64)
    assert meta[3] == len(mf) == len(exp["mem"]) > 0
    em = np.array(exp["mem"], dtype=np.int64)
    assert np.array_equal(order[mf[:, 0]], em[:, 0]) and np.array_equal(order[mf[:, 1]], em[:, 1])
    assert np.array_equal(mf[:, 2:5], em[:, 2:5]) and (mf[:, 5] == 1).all()
    # plus side = lower subdomain tag (ECS-like) wherever the tags differ
    tp, tm = ctags[em[:, 0]].astype(np.int64), ctags[em[:, 1]].astype(np.int64)
    assert (tp <= tm).all()
    if which == "2D_r0":
        assert (tp == tm).any()               # too coarse for the ICS box: membrane-tagged facets between EQUAL-tag cells exercise the n('-') rule
    if which == "emix":
        assert set(np.unique(tm)) == {1, 2} and (tp == 0).all()      # glial and neuronal cells always face the ECS

    # cells: the caller's local vertex order (it carries the facet matching and the DoF numbering dof(c, a) = c * nd + a); only the
    # vertex STORAGE ids are relabelled
    cells_d = dev.debug_table(A.DT_CELLS).reshape(nc, nv)
    assert np.array_equal(cells_d, dev.vertex_rank[mesh.cells[order]])

    # halo-block tables of the 3D P1 apply
    src, loc = dev.debug_table(A.DT_HB_SRC), dev.debug_table(A.DT_HB_LOC)
    if nv == 4 and loc.size:
        hs, long0 = int(meta[4]), int(meta[5])
        eloc, lists = _expected_halo_tables(nbr_d, fb & 3, (fb >> 2) & 3, nc)
        nblk = (nc + 255) // 256
        assert long0 == next((b for b in range(nblk) if len(lists[b]) > 256), nblk)
        assert hs == (max(len(l) for l in lists[:long0]) + 7) // 8 * 8
        assert np.array_equal(loc.reshape(nc, 4), eloc)
        src = src.reshape(nblk, hs)
        for b in range(long0):
            L = lists[b]
            assert np.array_equal(src[b, :len(L)], np.array(L, dtype=np.int32)) and (src[b, len(L):] == -1).all()
    else:
        assert which == "2D_r0" and src.size == 0 and loc.size == 0
    dev.close()


def test_dof_layout_upload_download_bit_exact(hip_lib):
    """DoF layout: dof(c, a) = c * nd + a in the caller's numbering on both sides of the boundary, species-major
    for the KNP fields; the device's Morton order is invisible: what is uploaded comes back bit for bit, and a cell-indexed
    pattern lands on the cell it names (checked through the operator-free facet trace: plus / minus traces of a field whose
    value is the caller cell index return the indices of the oracle's plus / minus cells)."""
    mesh, sub, surf, mtags = _mesh("3D_4axon_r0")
    ctags, ftags = sub.array(), surf.array()
    exp = oc.derive_tables(mesh.cells, ctags, mesh.facets, ftags, mtags)
    dev = A.Device(mesh, ctags, ftags, mtags, 3)
    nc, nd = mesh.num_cells(), 4
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, nc, nd))
    dev.upload(A.F_C, x)
    assert np.array_equal(dev.download(A.F_C).reshape(2, nc, nd), x)
    ids = np.repeat(np.arange(nc, dtype=np.float64), nd)
    dev.upload(A.F_PHI, ids)
    em = np.array(exp["mem"], dtype=np.int64)
    for side, col in ((0, 0), (1, 1)):
        tr = dev.facet_trace(A.F_PHI, 0, side)
        got = tr[em[:, 4]]
        assert np.array_equal(np.rint(got), em[:, col].astype(np.float64)) and np.abs(got - np.rint(got)).max() < 1e-9
    dev.close()
