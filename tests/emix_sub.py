"""A ~15 k-tet piece of the reference's EMIx tissue reconstruction (BASELINE configs[4]), small enough for the oracle's sparse
direct solves: the cells of `emix_common.load_mesh()` whose midpoint lies inside a fixed box, vertices renumbered monotonically
(cells keep ascending vertex ids = the shared indexing contract), subdomains inherited, facet tags re-derived from cell-label
disagreement exactly as for the full mesh (examples/emix_simulations/emix_common.py; reference:
examples/rat-neuron/run_rat_neuron.py:187-201).  Used by tests/golden/make_trajectories.py (oracle side) and
tests/test_gpu_trajectory.py (HIP side): data selection only, no arithmetic of either path."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))

# box in cm (the reconstruction spans about [0, 5e-4]^3 cm): straddles the stimulus plane x = 3e-4 cm
# (run_EMIx_simulation.py:152) and contains ECS, glial and neuronal cells with both membrane tags
BOX_LO = np.array([2.0e-4, 1.0e-4, 1.0e-4])
BOX_HI = np.array([4.0e-4, 3.2e-4, 3.2e-4])


def emix_submesh(lo=BOX_LO, hi=BOX_HI):
    from emix_common import MESH_XDMF, LABEL_TO_SUBDOMAIN
    from knpemidg.h5lite import read_xdmf_mesh
    from knpemidg.mesh import Mesh, MeshFunction
    coords, cells, attrs = read_xdmf_mesh(MESH_XDMF)
    coords = np.asarray(coords, dtype=np.float64) * 1e-7                     # nm -> cm
    cells = np.sort(np.asarray(cells, dtype=np.int64), axis=1)
    label = np.asarray(attrs["label"]).astype(np.int64)
    mid = coords[cells].mean(axis=1)
    keep = np.nonzero(((mid >= lo) & (mid <= hi)).all(axis=1))[0]
    # largest face-connected component of the selection (a cut leaves a few tets that touch the rest through an edge or a vertex
    # only: each would carry its own constant null vector of a_emi)
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    tmp = Mesh(coords, cells[keep])
    fc = tmp.facet_cells[tmp.facet_cells[:, 1] >= 0]
    graph = coo_matrix((np.ones(len(fc)), (fc[:, 0], fc[:, 1])), shape=(len(keep), len(keep)))
    _, comp = connected_components(graph, directed=False)
    keep = keep[comp == np.bincount(comp).argmax()]
    used = np.unique(cells[keep])
    new_id = np.full(coords.shape[0], -1, dtype=np.int64)
    new_id[used] = np.arange(len(used))
    mesh = Mesh(coords[used], new_id[cells[keep]])
    lab = label[keep]
    sub = np.vectorize(LABEL_TO_SUBDOMAIN.get)(lab).astype(np.uint32)
    fc = mesh.facet_cells
    interior = fc[:, 1] >= 0
    tags = np.zeros(mesh.num_facets(), dtype=np.uint32)
    tags[~interior] = 10
    l0, l1 = lab[fc[interior, 0]], lab[fc[interior, 1]]
    s0, s1 = sub[fc[interior, 0]], sub[fc[interior, 1]]
    t = np.zeros(int(interior.sum()), dtype=np.uint32)
    differ = l0 != l1
    t[differ] = np.where((s0 == 2) | (s1 == 2), 2, 1)[differ]
    tags[interior] = t
    return mesh, MeshFunction(mesh, 3, sub), MeshFunction(mesh, 2, tags)
