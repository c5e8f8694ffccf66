"""Array-backed stand-ins for dolfin Function / FunctionSpace objects that callers of the
reference `Solver` touch (attributes listed in SURVEY.md section 8b): `phi`, `c`, `c_prev_k`,
`c_prev_n`, `ion_list[-1]['c']`, `ion['E']`, `phi_M_prev_PDE`, `Q`."""
import numpy as np


class _Vector:
    """`.vector()` proxy with numpy-like access (get_local / set_local / [:])."""

    def __init__(self, getter, setter):
        self._get, self._set = getter, setter

    def get_local(self):
        return self._get()

    def set_local(self, a):
        self._set(np.asarray(a, dtype=np.float64))

    def __getitem__(self, i):
        return self._get()[i]

    def __setitem__(self, i, v):
        a = self._get()
        a[i] = v
        self._set(a)

    def __array__(self, dtype=None):
        return self._get()

    def __len__(self):
        return len(self._get())

    def __sub__(self, o):
        return self._get() - np.asarray(o)

    def apply(self, mode):
        pass


class FacetSpace:
    """DLT0 space `Q`: one value per facet (solver.py:209)."""

    def __init__(self, mesh):
        self._mesh = mesh

    def mesh(self):
        return self._mesh

    def dim(self):
        return self._mesh.num_facets()

    def tabulate_dof_coordinates(self):
        return self._mesh.facet_midpoints()


class FacetFunction:
    """Host-resident DLT0 function."""

    def __init__(self, Q, values=None):
        self.Q = Q
        self._a = np.zeros(Q.dim()) if values is None else np.asarray(values, dtype=np.float64).copy()

    def function_space(self):
        return self.Q

    def vector(self):
        return _Vector(lambda: self._a, lambda a: self._a.__setitem__(slice(None), a))

    def array(self):
        return self._a

    def assign(self, other):
        self._a[:] = other.array() if hasattr(other, "array") else np.asarray(other)


class DeviceFacetFunction(FacetFunction):
    """DLT0 function living in a device field row (PHI_M, E_k, I_ch_k, or a scratch slot holding a
    pcws_constant_project result -- then `generation` identifies the projection that wrote the slot)."""

    def __init__(self, Q, dev, field, row=0, generation=None):
        self.Q, self.dev, self.field, self.row, self.generation = Q, dev, field, row, generation

    def _n(self):
        return self.Q.dim()

    def check_valid(self):
        if self.generation is not None and not self.dev.tmp_slot_valid(self.row, self.generation):
            raise RuntimeError("this pcws_constant_project result has been overwritten by later projections "
                               "(scratch slots are recycled): consume or copy it earlier")

    def array(self):
        self.check_valid()
        return self.dev.download(self.field, self.row * self._n(), self._n())

    def vector(self):
        return _Vector(self.array, lambda a: self.dev.upload(self.field, a, self.row * self._n()))

    def assign(self, other):
        self.dev.upload(self.field, other.array() if hasattr(other, "array") else np.asarray(other),
                        self.row * self._n())


class DeviceFunction:
    """DG-p nodal function living in a device field.  `component` selects a species block of a
    species-major [n_sys][nc*nd] field; `n_comp` > 1 marks the mixed function itself."""

    def __init__(self, dev, field, nc, nd, n_comp=1, component=0):
        self.dev, self.field, self.nc, self.nd = dev, field, nc, nd
        self.n_comp, self.component = n_comp, component

    @property
    def ndof(self):
        return self.nc * self.nd

    def split(self, deepcopy=False):
        return tuple(DeviceFunction(self.dev, self.field, self.nc, self.nd, 1, k) for k in range(self.n_comp))

    def sub(self, k):
        return self.split()[k]

    def array(self):
        """[nc, nd] for a scalar function, [n_comp, nc, nd] for the mixed one."""
        if self.n_comp > 1:
            return self.dev.download(self.field).reshape(self.n_comp, self.nc, self.nd)
        return self.dev.download(self.field, self.component * self.ndof, self.ndof).reshape(self.nc, self.nd)

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if self.n_comp > 1:
            assert a.size == self.n_comp * self.ndof
            self.dev.upload(self.field, a)
        else:
            assert a.size == self.ndof
            self.dev.upload(self.field, a, self.component * self.ndof)

    def vector(self):
        return _Vector(lambda: self.array().ravel(), self.set)

    def assign(self, other):
        self.set(other.array() if hasattr(other, "array") else np.asarray(other))
