"""Array-backed stand-ins for the dolfin objects the reference `Solver` API takes.

The reference hands `Solver.setup_domain(mesh, subdomains, surfaces)` a
`dolfin.Mesh` and two `dolfin.MeshFunction`s (reference: src/knpemidg/solver.py:85-121).
dolfin cannot exist on the GPU box, so the drop-in boundary accepts these
numpy-backed equivalents instead:

* `Mesh`          coords f64[Nv,d], cells i32[Nc,d+1] (vertices sorted ascending per cell)
                  plus the facet table derived from them,
* `MeshFunction`  one unsigned tag per cell (dim == tdim) or per facet (dim == tdim-1),
* `Constant`      mutable scalar with `float()` / `.assign()` (needed for `t`,
                  reference: src/knpemidg/solver.py:845).

The structured generators restate the recipes of the reference's mesh scripts
(examples/idealized-geometries/make_mesh_2D.py:75-92, make_mesh_3D.py:81-111,
tests/make_mesh_MMS.py:64-102).  The vertex / cell numbering of
`RectangleMesh` / `BoxMesh` follows DOLFIN's generators as recalled (third-party,
not vendored in the reference tree); nothing downstream depends on that numbering
except memory locality.

Local facet `i` of a cell is the facet opposite local vertex `i` (UFC convention).
All integer tables here are the "bit-exact indexing" contract shared by the CPU
oracle and the HIP path.
"""
import numpy as np


class Constant:
    """Tiny mutable scalar (dolfin.Constant stand-in)."""

    def __init__(self, value):
        self._v = float(value)

    def assign(self, value):
        self._v = float(value)

    def __float__(self):
        return self._v

    def __add__(self, o):
        return self._v + float(o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._v - float(o)

    def __rsub__(self, o):
        return float(o) - self._v

    def __mul__(self, o):
        return self._v * float(o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._v / float(o)

    def __rtruediv__(self, o):
        return float(o) / self._v

    def __neg__(self):
        return -self._v

    def __repr__(self):
        return "Constant(%r)" % self._v


def _as_float(v):
    return float(v)


class Mesh:
    """Simplicial mesh + facet table.

    Attributes
    ----------
    coords        f64[Nv, d]
    cells         i32[Nc, d+1]   vertex ids, ascending per cell
    facets        i32[Nf, d]     vertex ids, ascending per facet
    facet_cells   i32[Nf, 2]     the two cells sharing the facet; [:,1] == -1 on the boundary.
                                 Side 0 is the cell with the lower index.
    facet_local   i8 [Nf, 2]     local facet index of the facet in each of those cells (-1 if none)
    cell_facets   i32[Nc, d+1]   global facet id of local facet i of each cell
    """

    def __init__(self, coords, cells):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        cells = np.asarray(cells)
        assert coords.ndim == 2 and cells.ndim == 2
        assert cells.shape[1] == coords.shape[1] + 1, "simplicial meshes only"
        if not (cells.dtype == np.int32 and (cells[:, 1:] > cells[:, :-1]).all()):      # generators hand over ascending int32 rows already
            cells = np.sort(cells.astype(np.int64), axis=1).astype(np.int32)
        self.coords = coords
        self.cells = np.ascontiguousarray(cells)
        self.gdim = coords.shape[1]
        self._build_facets()

    # -- dolfin-like accessors -------------------------------------------------
    def num_cells(self):
        return self.cells.shape[0]

    def num_vertices(self):
        return self.coords.shape[0]

    def num_facets(self):
        return self.facets.shape[0]

    def coordinates(self):
        return self.coords

    class _Geom:
        def __init__(self, d):
            self._d = d

        def dim(self):
            return self._d

    def geometry(self):
        return Mesh._Geom(self.gdim)

    def topology(self):
        return Mesh._Geom(self.gdim)

    # -- tables ----------------------------------------------------------------
    def _build_facets(self):
        if self.cells.shape[0] >= 20000 and self._build_facets_native():
            return
        cells = self.cells.astype(np.int64)
        nc, nv = cells.shape
        d = nv - 1
        # (cell, local facet) -> sorted vertex tuple of the facet opposite vertex i
        keys = np.empty((nc, nv, d), dtype=np.int64)
        for i in range(nv):
            keys[:, i, :] = np.delete(cells, i, axis=1)  # stays ascending
        flat = keys.reshape(nc * nv, d)
        order = np.lexsort(tuple(flat[:, j] for j in range(d - 1, -1, -1)))
        srt = flat[order]
        new = np.ones(len(srt), dtype=bool)
        new[1:] = np.any(srt[1:] != srt[:-1], axis=1)
        # number facets by first appearance in (cell, local facet) order so that
        # facet ids follow the cell ordering (memory locality of facet fields)
        grp = np.cumsum(new) - 1                       # group id in sorted order
        ngrp = grp[-1] + 1
        first = np.full(ngrp, nc * nv, dtype=np.int64)
        np.minimum.at(first, grp, order)
        rank = np.empty(ngrp, dtype=np.int64)
        rank[np.argsort(first, kind="stable")] = np.arange(ngrp)
        fid_sorted = rank[grp]
        fid = np.empty(nc * nv, dtype=np.int64)
        fid[order] = fid_sorted
        nf = ngrp
        self.cell_facets = np.ascontiguousarray(fid.reshape(nc, nv).astype(np.int32))
        self.facets = np.empty((nf, d), dtype=np.int32)
        self.facets[fid] = flat
        fc = np.full((nf, 2), -1, dtype=np.int32)
        fl = np.full((nf, 2), -1, dtype=np.int8)
        cidx = np.repeat(np.arange(nc), nv)
        lidx = np.tile(np.arange(nv), nc)
        # cells visited in ascending order: first hit fills side 0, second side 1
        cnt = np.zeros(nf, dtype=np.int64)
        o2 = np.argsort(fid, kind="stable")
        f_s, c_s, l_s = fid[o2], cidx[o2], lidx[o2]
        firstocc = np.ones(len(f_s), dtype=bool)
        firstocc[1:] = f_s[1:] != f_s[:-1]
        fc[f_s[firstocc], 0] = c_s[firstocc]
        fl[f_s[firstocc], 0] = l_s[firstocc]
        fc[f_s[~firstocc], 1] = c_s[~firstocc]
        fl[f_s[~firstocc], 1] = l_s[~firstocc]
        np.add.at(cnt, fid, 1)
        assert cnt.max() <= 2, "non-manifold mesh"
        self.facet_cells = fc
        self.facet_local = fl

    def _build_facets_native(self):
        """The same tables from the library's hash-based builder (csrc/host_sparse.cpp: knp_host_build_facets; identical numbering: facets
        by first appearance in (cell, local facet) order) -- the numpy version below sorts 4 nc keys three times (0.4 s at 10^6 tets)."""
        try:
            from knpemidg import _abi
            lib = _abi.load()
        except Exception:
            return False
        nc, nv = self.cells.shape
        cells = np.ascontiguousarray(self.cells, dtype=np.int32)
        cf = np.empty((nc, nv), dtype=np.int32)
        facets = np.empty((nc * nv, nv - 1), dtype=np.int32)
        fc = np.empty((nc * nv, 2), dtype=np.int32)
        fl = np.empty((nc * nv, 2), dtype=np.int8)
        nf = int(lib.knp_host_build_facets(nc, nv, _abi._p(cells, _abi._i32p), _abi._p(cf, _abi._i32p), _abi._p(facets, _abi._i32p),
                                           _abi._p(fc, _abi._i32p), _abi._p(fl, _abi._i8p)))
        if nf == -2:
            raise AssertionError("non-manifold mesh")
        if nf < 0:
            return False
        self.cell_facets = cf
        self.facets = np.ascontiguousarray(facets[:nf])
        self.facet_cells = np.ascontiguousarray(fc[:nf])
        self.facet_local = np.ascontiguousarray(fl[:nf])
        return True

    def _midpoints(self, what, conn):
        """Midpoints of the cells / facets, computed once (mesh builders, the Morton order and the partitioner all ask for them);
        invalidated when the coordinate array is replaced or rescaled in place."""
        stamp = (id(self.coords), float(self.coords[0, 0]), float(self.coords[-1, -1]))
        cache = self.__dict__.setdefault("_midpoint_cache", {})
        hit = cache.get(what)
        if hit is None or hit[0] != stamp:
            acc = self.coords[conn[:, 0]].copy()
            for k in range(1, conn.shape[1]):
                acc += self.coords[conn[:, k]]
            acc /= conn.shape[1]
            acc.setflags(write=False)
            hit = cache[what] = (stamp, acc)
        return hit[1]

    def facet_midpoints(self):
        return self._midpoints("facets", self.facets)

    def cell_midpoints(self):
        return self._midpoints("cells", self.cells)

    def interior_facets(self):
        return np.nonzero(self.facet_cells[:, 1] >= 0)[0]

    def exterior_facets(self):
        return np.nonzero(self.facet_cells[:, 1] < 0)[0]

    def hmin(self):
        """Smallest cell diameter (dolfin `mesh.hmin()` uses 2*circumradius; the
        reference only uses it for convergence-rate bookkeeping, tests/run_MMS_space.py:264)."""
        return float(cell_diameters(self).min())


def cell_diameters(mesh):
    """UFL `CellDiameter`: largest vertex-to-vertex distance of each cell
    (reference: src/knpemidg/solver.py:102-103; UFL semantics recalled, third-party)."""
    x = mesh.coords[mesh.cells]                       # [Nc, d+1, d]
    nv = x.shape[1]
    h = np.zeros(x.shape[0])
    for a in range(nv):
        for b in range(a + 1, nv):
            h = np.maximum(h, np.linalg.norm(x[:, a] - x[:, b], axis=1))
    return h


def refine_uniform(mesh):
    """Regular (red) refinement of a tetrahedral mesh: every tet -> 8 (four corner tets + the inner octahedron cut along the
    diagonal m02 - m13, Bey's rule), every edge gets one midpoint vertex.  Returns (refined Mesh, parent cell of every new cell).
    Old vertices keep their ids.  Used to bring the reference's bundled EMIx reconstruction (121 617 tets) to a size at which its
    unstructured kernels are measured outside the launch-latency regime (bench.py --workload emix --refine 1: 972 936 tets)."""
    assert mesh.gdim == 3
    cells = mesh.cells.astype(np.int64)
    nv = mesh.coords.shape[0]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    e = np.stack([np.stack([cells[:, a], cells[:, b]], axis=1) for a, b in pairs], axis=1)       # [Nc, 6, 2], ascending ids
    key = e[:, :, 0] * nv + e[:, :, 1]
    uniq, inv = np.unique(key.ravel(), return_inverse=True)
    mid = (nv + inv).reshape(-1, 6)                                                              # midpoint vertex of each cell edge
    coords = np.concatenate([mesh.coords, 0.5 * (mesh.coords[uniq // nv] + mesh.coords[uniq % nv])])
    v0, v1, v2, v3 = cells.T
    m01, m02, m03, m12, m13, m23 = mid.T
    kids = [(v0, m01, m02, m03), (v1, m01, m12, m13), (v2, m02, m12, m23), (v3, m03, m13, m23),
            (m01, m02, m03, m13), (m01, m02, m12, m13), (m02, m03, m13, m23), (m02, m12, m13, m23)]
    new_cells = np.stack([np.stack(k, axis=1) for k in kids], axis=1).reshape(-1, 4)
    parent = np.repeat(np.arange(cells.shape[0]), 8)
    return Mesh(coords, new_cells), parent


class MeshFunction:
    """One unsigned tag per mesh entity of dimension `dim` (dolfin.MeshFunction('size_t') stand-in)."""

    def __init__(self, mesh, dim, value=0):
        self._mesh = mesh
        self._dim = int(dim)
        tdim = mesh.gdim
        if self._dim == tdim:
            n = mesh.num_cells()
        elif self._dim == tdim - 1:
            n = mesh.num_facets()
        else:
            raise ValueError("MeshFunction supports cells and facets only")
        if np.ndim(value) == 0:
            self._a = np.full(n, int(value), dtype=np.uint32)
        else:
            a = np.asarray(value)
            assert a.shape == (n,)
            self._a = np.ascontiguousarray(a.astype(np.uint32))

    def mesh(self):
        return self._mesh

    def dim(self):
        return self._dim

    def array(self):
        return self._a

    def where_equal(self, tag):
        return np.nonzero(self._a == int(tag))[0]

    def __getitem__(self, i):
        return self._a[i]

    def __setitem__(self, i, v):
        self._a[i] = v

    def __len__(self):
        return len(self._a)


# --------------------------------------------------------------------------
# structured generators (DOLFIN numbering as recalled)
# --------------------------------------------------------------------------
def RectangleMesh(p0, p1, nx, ny, diagonal="right"):
    x0, y0 = float(p0[0]), float(p0[1])
    x1, y1 = float(p1[0]), float(p1[1])
    xs = x0 + (x1 - x0) * np.arange(nx + 1) / nx
    ys = y0 + (y1 - y0) * np.arange(ny + 1) / ny
    X, Y = np.meshgrid(xs, ys, indexing="xy")          # [ny+1, nx+1], x fastest
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    ix = ix.ravel()
    iy = iy.ravel()
    v0 = iy * (nx + 1) + ix
    v1 = v0 + 1
    v2 = v0 + (nx + 1)
    v3 = v1 + (nx + 1)
    if diagonal == "crossed":
        xm = 0.5 * (xs[:-1] + xs[1:])
        ym = 0.5 * (ys[:-1] + ys[1:])
        XM, YM = np.meshgrid(xm, ym, indexing="xy")
        coords = np.vstack([coords, np.stack([XM.ravel(), YM.ravel()], axis=1)])
        vm = (nx + 1) * (ny + 1) + iy * nx + ix
        tris = np.stack([np.stack([v0, v1, vm], 1), np.stack([v0, v2, vm], 1),
                         np.stack([v1, v3, vm], 1), np.stack([v2, v3, vm], 1)], axis=1)
    elif diagonal == "right":
        tris = np.stack([np.stack([v0, v1, v3], 1), np.stack([v0, v2, v3], 1)], axis=1)
    elif diagonal == "left":
        tris = np.stack([np.stack([v0, v1, v2], 1), np.stack([v1, v2, v3], 1)], axis=1)
    else:
        raise ValueError("diagonal must be 'right', 'left' or 'crossed'")
    return Mesh(coords, tris.reshape(-1, 3))


def BoxMesh(p0, p1, nx, ny, nz):
    p0 = np.asarray(p0, dtype=float)
    p1 = np.asarray(p1, dtype=float)
    xs = p0[0] + (p1[0] - p0[0]) * np.arange(nx + 1) / nx
    ys = p0[1] + (p1[1] - p0[1]) * np.arange(ny + 1) / ny
    zs = p0[2] + (p1[2] - p0[2]) * np.arange(nz + 1) / nz
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")   # x fastest
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    itype = np.int32 if (nx + 1) * (ny + 1) * (nz + 1) < 2 ** 31 else np.int64
    iz, iy, ix = np.meshgrid(np.arange(nz, dtype=itype), np.arange(ny, dtype=itype), np.arange(nx, dtype=itype), indexing="ij")
    sx, sy = nx + 1, (nx + 1) * (ny + 1)
    v0 = (iz * sy + iy * sx + ix).ravel()
    # the six tetrahedra of a brick, (v0, v1, v3, v7) (v0, v1, v7, v5) (v0, v5, v7, v4) (v0, v3, v2, v7) (v0, v6, v4, v7) (v0, v2, v6, v7) with
    # v1 = v0 + 1, v2 = v0 + sx, v3 = v1 + sx, v4 .. v7 = v0 .. v3 + sy, written with every row already ascending (what Mesh() sorts them to)
    off = np.array([[0, 1, 1 + sx, 1 + sx + sy], [0, 1, 1 + sy, 1 + sx + sy], [0, sy, 1 + sy, 1 + sx + sy],
                    [0, sx, 1 + sx, 1 + sx + sy], [0, sy, sx + sy, 1 + sx + sy], [0, sx, sx + sy, 1 + sx + sy]], dtype=itype)
    tets = v0[:, None, None] + off[None, :, :]
    return Mesh(coords, tets.reshape(-1, 4))


def _near(a, b, eps=3.0e-16):
    # dolfin.near(): |a-b| < DOLFIN_EPS (absolute)
    return np.abs(a - b) < eps


def _box_marks_native(mesh, a, b, eps):
    """(cells inside the box, facets on its surface) through the library (csrc/host_sparse.cpp: knp_host_box_marks; the same tests on the
    same midpoints, one threaded pass each instead of ~30 numpy passes over the facet midpoints).  (None, None) without the library."""
    try:
        from knpemidg import _abi
        lib = _abi.load()
    except (OSError, ImportError):
        return None, None
    co = np.ascontiguousarray(mesh.coords, dtype=np.float64)
    u8p = _abi.C.POINTER(_abi.C.c_uint8)
    out = []
    for conn, mode in ((mesh.cells, 0), (mesh.facets, 1)):
        cn = np.ascontiguousarray(conn, dtype=np.int32)
        m = np.empty(cn.shape[0], dtype=np.uint8)
        rc = lib.knp_host_box_marks(cn.shape[0], mesh.gdim, _abi._p(co, _abi._f64p), _abi._p(cn, _abi._i32p), cn.shape[1], _abi._p(a, _abi._f64p),
                                    _abi._p(b, _abi._f64p), float(eps), mode, _abi._p(m, u8p), 0)
        if rc != 0:
            return None, None
        out.append(m.view(np.bool_))
    return out[0], out[1]


def _tag_box(mesh, subdomains, surfaces, a, b, tag_bndr, additive=False, eps=1e-12):
    """Mark cells whose midpoint lies in [a,b] with 1 and the facets on the box
    surface with `tag_bndr` (reference: make_mesh_3D.py:15-50, make_mesh_2D.py:24-47).
    `eps` absorbs the rounding of grid coordinates that dolfin's `near` absorbs."""
    d = mesh.gdim
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    inside = on = None
    if mesh.num_cells() >= 50000:
        inside, on = _box_marks_native(mesh, a, b, eps)
    if inside is None:
        cm = mesh.cell_midpoints()
        inside = np.all((cm >= a) & (cm <= b), axis=1)
        fm = mesh.facet_midpoints()
        on = np.zeros(len(fm), dtype=bool)
        for ax in range(d):
            others = [o for o in range(d) if o != ax]
            within = np.ones(len(fm), dtype=bool)
            for o in others:
                within &= (fm[:, o] >= a[o] - eps) & (fm[:, o] <= b[o] + eps)
            on |= within & (np.abs(fm[:, ax] - a[ax]) < eps)
            on |= within & (np.abs(fm[:, ax] - b[ax]) < eps)
    subdomains.array()[inside] = 1
    assert inside.any()
    if additive:
        surfaces.array()[on] += 1
    else:
        surfaces.array()[on] = tag_bndr
    return inside, on


def make_mesh_2D(resolution_factor=0):
    """2D single neuron in ECS (reference: examples/idealized-geometries/make_mesh_2D.py:75-92).
    Returns (mesh, subdomains, surfaces); lengths in metres."""
    nx = 31 * 2 ** resolution_factor
    ny = 2 * 2 ** resolution_factor
    mesh = RectangleMesh((0, 0), (62, 4), nx, ny, "crossed")
    subdomains = MeshFunction(mesh, 2, 0)
    surfaces = MeshFunction(mesh, 1, 0)
    _tag_box(mesh, subdomains, surfaces, (1, 1), (61, 3), 1, additive=True)
    surfaces.array()[mesh.exterior_facets()] = 5
    mesh.coords *= 1e-6
    return mesh, subdomains, surfaces


def make_mesh_3D(resolution_factor=0, n_axons=4):
    """3D box with up to 4 axons (reference: examples/idealized-geometries/make_mesh_3D.py:81-111).
    `n_axons=1` keeps only the first axon (BASELINE config 2, "3D idealized single cell")."""
    l = 2
    nx = l * 16 * 2 ** resolution_factor
    ny = 9 * 2 ** resolution_factor
    nz = 9 * 2 ** resolution_factor
    mesh = BoxMesh((0, 0.0, 0.0), (l * 16, 0.9, 0.9), nx, ny, nz)
    subdomains = MeshFunction(mesh, 3, 0)
    surfaces = MeshFunction(mesh, 2, 0)
    axons = [((5, 0.2, 0.2), (l * 16 - 5, 0.4, 0.4), 1),
             ((5, 0.5, 0.5), (l * 16 - 5, 0.7, 0.7), 2),
             ((5, 0.5, 0.2), (l * 16 - 5, 0.7, 0.4), 2),
             ((5, 0.2, 0.5), (l * 16 - 5, 0.4, 0.7), 2)]
    for a, b, tag in axons[:n_axons]:
        _tag_box(mesh, subdomains, surfaces, a, b, tag)
    surfaces.array()[mesh.exterior_facets()] = 5
    mesh.coords *= 1e-6
    return mesh, subdomains, surfaces


def make_mesh_MMS(resolution_factor=4):
    """Unit square with ICS = [0.25,0.75]^2 and four interface tags
    (reference: tests/make_mesh_MMS.py:64-102)."""
    n = 2 ** resolution_factor
    mesh = RectangleMesh((0, 0), (1, 1), n, n, "right")
    subdomains = MeshFunction(mesh, 2, 0)
    surfaces = MeshFunction(mesh, 1, 0)
    a, b = (0.25, 0.25), (0.75, 0.75)
    cm = mesh.cell_midpoints()
    inside = np.all((cm >= a) & (cm <= b), axis=1)
    subdomains.array()[inside] = 1
    fm = mesh.facet_midpoints()
    eps = 1e-12
    iny = (fm[:, 1] >= a[1]) & (fm[:, 1] <= b[1])
    inx = (fm[:, 0] >= a[0]) & (fm[:, 0] <= b[0])
    s1 = (np.abs(fm[:, 0] - a[0]) < eps) & iny
    s2 = (np.abs(fm[:, 1] - a[1]) < eps) & inx
    s3 = (np.abs(fm[:, 0] - b[0]) < eps) & iny
    s4 = (np.abs(fm[:, 1] - b[1]) < eps) & inx
    tags = np.where(s1, 1, np.where(s2, 2, np.where(s3, 3, np.where(s4, 4, 0))))
    surfaces.array()[:] = tags
    ext = mesh.exterior_facets()
    fe = fm[ext]
    surfaces.array()[ext[np.abs(fe[:, 0] - 0.0) < eps]] = 5
    surfaces.array()[ext[np.abs(fe[:, 1] - 0.0) < eps]] = 6
    surfaces.array()[ext[np.abs(fe[:, 0] - 1.0) < eps]] = 7
    surfaces.array()[ext[np.abs(fe[:, 1] - 1.0) < eps]] = 8
    return mesh, subdomains, surfaces
