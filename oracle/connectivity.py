"""TEST INFRASTRUCTURE (oracle) -- connectivity / orientation tables derived independently of the product.

Only tests/ may import this module.  It restates, from raw arrays only (cell -> vertex ids, cell tags, a tag per facet keyed
by the facet's vertex tuple), what the reference obtains from DOLFIN's topology and from its tag logic:

* which cell lies behind every local facet of every cell (DOLFIN `mesh.topology()(D-1, D)`; a tetrahedron's local facet i is the
  one opposite local vertex i, as in UFC),
* the integration class of every interior facet: `dS(0)` = SIPG coupling, `dS(tag)` for the membrane tags of the model
  dictionary = membrane coupling, any other tag = no coupling (reference: src/knpemidg/solver.py:113-121, 325-345, 586-627),
* the orientation of the interface normal `n_g` and with it the `plus` / `minus` restrictions (reference:
  src/knpemidg/utils.py:61-98): n_g leaves the cell with the LOWER tag; on equal tags the reference takes n('-'), and this
  build fixes '+' = the cell with the lower index (caller numbering), so on equal tags `plus` = the higher-index cell.

Nothing here uses knpemidg.mesh's facet tables (facet_cells / facet_local / cell_facets): faces are matched through a Python
dictionary on sorted vertex tuples.  PARITY UNPINNED: the reference holds no connectivity fixtures.
"""
import numpy as np

K_SIPG, K_MEMBRANE, K_EXTERIOR, K_INACTIVE = 0, 1, 2, 3


def derive_tables(cells, cell_tags, facet_vertices, facet_tags, membrane_tags):
    """cells [nc, nv] (ascending vertex ids per cell), cell_tags [nc], facet_vertices [nf, nv-1] + facet_tags [nf] (the caller's
    facet numbering and tags), membrane_tags: iterable.

    Returns a dict of caller-numbered tables:
      nbr [nc, nv]      cell behind local facet i, -1 on the boundary
      nloc [nc, nv]     local facet index of that facet in the neighbour (0 on the boundary)
      kind [nc, nv]     K_* class
      plus [nc, nv]     1 if this cell is the `plus` side of the facet (0 on the boundary)
      fid [nc, nv]      caller's facet id
      mem               list of (plus cell, minus cell, local facet in plus, local facet in minus, facet id), facet id ascending
    """
    cells = np.asarray(cells)
    nc, nv = cells.shape
    tag_of = {}
    fid_of = {}
    for f, (vs, t) in enumerate(zip(np.asarray(facet_vertices).tolist(), np.asarray(facet_tags).tolist())):
        key = tuple(sorted(vs))
        tag_of[key] = int(t)
        fid_of[key] = f
    mset = set(int(t) for t in membrane_tags)
    seen = {}
    nbr = np.full((nc, nv), -1, dtype=np.int64)
    nloc = np.zeros((nc, nv), dtype=np.int64)
    kind = np.full((nc, nv), K_EXTERIOR, dtype=np.int64)
    plus = np.zeros((nc, nv), dtype=np.int64)
    fid = np.full((nc, nv), -1, dtype=np.int64)
    ctag = np.asarray(cell_tags).astype(np.int64)
    mem = []
    for c, vs in enumerate(cells.tolist()):
        for i in range(nv):
            key = tuple(sorted(vs[:i] + vs[i + 1:]))
            fid[c, i] = fid_of[key]
            other = seen.pop(key, None)
            if other is None:
                seen[key] = (c, i)
                continue
            c0, i0 = other                      # c0 < c: cells are visited in ascending order
            nbr[c, i], nloc[c, i] = c0, i0
            nbr[c0, i0], nloc[c0, i0] = c, i
            t = tag_of[key]
            k = K_SIPG if t == 0 else (K_MEMBRANE if t in mset else K_INACTIVE)
            kind[c, i] = kind[c0, i0] = k
            if ctag[c0] < ctag[c]:
                p = (c0, i0, c, i)
            elif ctag[c0] > ctag[c]:
                p = (c, i, c0, i0)
            else:
                p = (c, i, c0, i0)              # equal tags: n('-') with '+' = the lower-index cell -> plus = the higher-index cell
            plus[p[0], p[1]] = 1
            if k == K_MEMBRANE:
                mem.append((p[0], p[2], p[1], p[3], fid_of[key]))
    mem.sort(key=lambda r: r[4])
    return dict(nbr=nbr, nloc=nloc, kind=kind, plus=plus, fid=fid, mem=mem)
