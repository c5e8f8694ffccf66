"""Array-backed equivalents of knpemidg.utils (reference: src/knpemidg/utils.py).

* `subdomain_marking_foo`  cell tags as a DG0 array (the reference's only native code, the
  `PCWS` C++ Expression utils.py:5-39, is "read cell_tag[c]").
* `interface_normal`       oriented interface normal n_g (utils.py:61-85): for every interior
  facet, which side the normal LEAVES (`plus`, lower tag) and the unit vector itself.
* `plus` / `minus`         lazy trace expressions (utils.py:87-98).
* `pcws_constant_project`  facet average onto the DLT0 space Q (utils.py:100-124); evaluated on
  the device for nodal device fields (knp_facet_trace), membrane facets only.
"""
import numpy as np


def subdomain_marking_foo(subdomains, V=None):
    """DG0 function whose cell values are the cell tags (utils.py:44-59)."""
    return np.asarray(subdomains.array(), dtype=np.float64).copy()


class InterfaceNormal:
    """n_g: `plus_side[f]` in {0,1} is the facet side (index into Mesh.facet_cells) that the normal
    leaves (lower tag; side 1 on equal tags, matching the reference's `n('-')` choice with this
    build's '+' = side 0); `vector[f]` is the unit normal itself (outward on exterior facets)."""

    def __init__(self, subdomains, mesh):
        tags = np.asarray(subdomains.array() if hasattr(subdomains, "array") else subdomains).astype(np.int64)
        fc = mesh.facet_cells
        interior = fc[:, 1] >= 0
        plus = np.full(fc.shape[0], -1, dtype=np.int8)
        t0 = tags[fc[interior, 0]]
        t1 = tags[fc[interior, 1]]
        plus[interior] = np.where(t0 >= t1, 1, 0)
        self.plus_side = plus
        self.mesh = mesh
        self._vector = None

    @property
    def vector(self):
        if self._vector is None:
            m = self.mesh
            d = m.gdim
            fx = m.coords[m.facets]
            if d == 2:
                t = fx[:, 1] - fx[:, 0]
                n = np.stack([t[:, 1], -t[:, 0]], axis=1)
            else:
                n = np.cross(fx[:, 1] - fx[:, 0], fx[:, 2] - fx[:, 0])
            n /= np.linalg.norm(n, axis=1)[:, None]
            # outward from side 0: away from that cell's vertex opposite the facet
            c0 = m.facet_cells[:, 0]
            l0 = m.facet_local[:, 0].astype(np.int64)
            apex = m.coords[m.cells[c0, l0]]
            s = np.sign(np.einsum("fd,fd->f", n, fx[:, 0] - apex))
            n *= s[:, None]
            # lower -> higher tag: leaves plus side
            flip = self.plus_side == 1
            n[flip] *= -1.0
            self._vector = n
        return self._vector


def interface_normal(subdomains, mesh):
    return InterfaceNormal(subdomains, mesh)


class Trace:
    """`plus(f, n_g)` / `minus(f, n_g)`: restriction of a nodal field to one side of n_g."""

    def __init__(self, f, normal, side):
        self.f, self.normal, self.side = f, normal, side


def plus(phi, normal):
    """Trace on the cell the normal originates from (utils.py:87-91)."""
    return Trace(phi, normal, 0)


def minus(phi, normal):
    """Trace on the cell at which the normal ends (utils.py:94-98)."""
    return Trace(phi, normal, 1)


def pcws_constant_project(f, V, fV=None):
    """Facet average of a trace expression onto the DLT0 space `V` (utils.py:100-124).
    Supported expressions: plus(u)/minus(u) of a nodal device field (what every `update_ode`
    hook of the reference uses, e.g. run_3D.py:44-49).  Values are defined on membrane facets."""
    from knpemidg.functions import FacetFunction, DeviceFunction
    if not isinstance(f, Trace):
        raise NotImplementedError("pcws_constant_project supports plus(u, n_g) / minus(u, n_g) expressions")
    u = f.f
    if not isinstance(u, DeviceFunction):
        raise NotImplementedError("pcws_constant_project expects a device-backed nodal function")
    from knpemidg import _abi
    from knpemidg.functions import DeviceFacetFunction
    slot, gen = u.dev.facet_trace(u.field, u.component, f.side, download=False)
    # the result lives in one of the device's scratch facet slots (recycled after a few further projections; a stale
    # result raises instead of silently aliasing a later one)
    out = DeviceFacetFunction(V, u.dev, _abi.F_FACET_TMP, row=slot, generation=gen)
    if fV is not None:
        fV.assign(out)
        return fV
    return out


def CellCenterDistance(mesh):
    """Cell-centre distance per facet (utils.py:126-164; unused by Solver)."""
    cm = mesh.cell_midpoints()
    fm = mesh.facet_midpoints()
    fc = mesh.facet_cells
    out = np.linalg.norm(cm[fc[:, 0]] - fm, axis=1)
    it = fc[:, 1] >= 0
    out[it] = np.linalg.norm(cm[fc[it, 0]] - cm[fc[it, 1]], axis=1)
    return out
