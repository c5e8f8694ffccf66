"""`MembraneModel`: one ODE system per membrane facet (reference: src/knpemidg/membrane.py:7-186).

Same constructor and method names as the reference.  Differences forced by the environment
(SURVEY.md section 8b/8f-1): the ODE module supplies a vectorised `rhs(t, states, parameters)`
(numpy, all nodes at once) instead of a numba cfunc address, and `step_lsoda` integrates the
whole batch with an adaptive Dormand-Prince 5(4) pair under the reference's tolerance
(rtol 1e-8, membrane.py:112) instead of one numbalsoda call per facet.  The channel currents
`I_ch_k` are evaluated at the END state (the reference leaves whatever LSODA's last internal RHS
call wrote, mm_hh.py:154-159), so ODE outputs agree at tolerance level, never bitwise.
"""
import numpy as np

from knpemidg.functions import FacetFunction


def is_dlt_scalar(V):
    return hasattr(V, "tabulate_dof_coordinates")


def get_indices(V, facet_f, tags):
    """Facets (= DLT0 dofs) carrying one of `tags` (reference: dlt_dof_extraction.py:18-48)."""
    marked = np.sort(np.unique(np.concatenate([np.nonzero(facet_f.array() == t)[0] for t in tags])))
    return marked, marked.reshape(-1, 1)


def get_values(u, indices):
    return u.array()[np.asarray(indices).ravel()]


def set_values(u, indices, values):
    a = u.array().copy()
    a[np.asarray(indices).ravel()] = values
    u.vector().set_local(a)


# Dormand-Prince 5(4) tableau
_C = np.array([0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1, 1])
_A = [
    [],
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_B5 = np.array([35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0])
_B4 = np.array([5179 / 57600, 0, 7571 / 16695, 393 / 640, -92097 / 339200, 187 / 2100, 1 / 40])


def integrate_batch(rhs, t0, t1, y, params, rtol=1.0e-8, atol=1.0e-12, h0=None, max_steps=100000):
    """Advance all rows of `y` from t0 to t1.  Every row carries its OWN time and adaptive step, so a
    row's result does not depend on which other rows share the batch (ranks that both hold a membrane
    facet on a partition boundary therefore compute bitwise identical ODE outputs)."""
    n = y.shape[0]
    y = y.copy()
    t = np.full(n, float(t0))
    h = np.full(n, (t1 - t0) / 16) if h0 is None else np.array(h0, dtype=np.float64, copy=True)
    if h.shape != (n,):
        h = np.full(n, (t1 - t0) / 16)
    k = [None] * 7
    k[0] = rhs(t, y, params)
    done = np.zeros(n, dtype=bool)
    tiny = 1e-14 * max(abs(t1), 1e-30)
    steps = 0
    while not done.all() and steps < max_steps:
        hh = np.where(done, 0.0, np.minimum(h, t1 - t))
        hc = hh[:, None]
        for s in range(1, 7):
            ys = y + hc * sum(a * k[j] for j, a in enumerate(_A[s]) if a != 0)
            k[s] = rhs(t + _C[s] * hh, ys, params)
        y5 = y + hc * sum(b * kk for b, kk in zip(_B5, k) if b != 0)
        err = hc * sum((b5 - b4) * kk for b5, b4, kk in zip(_B5, _B4, k))
        scale = np.maximum(atol + rtol * np.maximum(np.abs(y), np.abs(y5)), 1e-300)
        e = np.max(np.abs(err) / scale, axis=1)
        e = np.where(np.isfinite(e), e, 1e10)
        steps += 1
        acc = ((e <= 1.0) | (hh < tiny)) & ~done
        t = np.where(acc, t + hh, t)
        y = np.where(acc[:, None], y5, y)
        k[0] = np.where(acc[:, None], k[6], k[0])          # FSAL for accepted rows
        done |= acc & (t >= t1 - 1e-15 * abs(t1))
        fac = np.clip(0.9 * (1.0 / np.maximum(e, 1e-10)) ** 0.2, 0.2, 5.0)
        h = np.where(done, h, hh * fac)
    if steps >= max_steps:
        raise AssertionError("ODE integrator did not reach the end time")   # `assert success`, membrane.py:113
    rhs(np.full(n, float(t1)), y, params)                 # leave I_ch_k evaluated at the end state
    return y, h


def _EVERYWHERE(x):
    return True


class MembraneModel:
    """ODE on the membrane facets where facet_f == tag (reference: membrane.py:7-41)."""

    def __init__(self, ode, facet_f, tag, V):
        mesh = facet_f.mesh()
        assert mesh.gdim - 1 == facet_f.dim()
        assert isinstance(tag, int)
        assert is_dlt_scalar(V)
        self.V = V
        self.facets, indices = get_indices(V, facet_f, (tag,))
        self.indices = indices.flatten()
        self.dof_locations = V.tabulate_dof_coordinates()[self.indices]
        nodes = len(self.indices)
        self.nodes = nodes
        s0 = np.asarray(ode.init_state_values(), dtype=np.float64)
        p0 = np.asarray(ode.init_parameter_values(), dtype=np.float64)
        self._states = np.tile(s0, (nodes, 1))               # [nodes, n_states]; well-formed for nodes == 0 too
        self._parameters = np.tile(p0, (nodes, 1))           # (a rank / tag without membrane facets)
        self.tag = tag
        self.ode = ode
        self.prefix = getattr(ode, "__name__", "ode")
        self.time = 0
        self._h = None
        self._dev = None
        self._handle = None
        self._stim_applied = None
        self._stim_locator = None
        self._stim_mask = None

    # --- device backing: tables live on the GPU, the batched HIP integrator steps them (csrc/ode.hip) ---
    def attach_device(self, dev):
        """Move the ODE tables to the device if the model has a device implementation (ode.MODEL_ID)."""
        if getattr(self.ode, "MODEL_ID", None) is None:
            return False
        self._handle = dev.ode_create(self.ode.MODEL_ID, self.indices, self._states, self._parameters)
        self._dev = dev
        return True

    @property
    def on_device(self):
        return self._dev is not None

    @property
    def states(self):
        """ODE states [nodes, n_states] (a snapshot when device backed; assign the whole array to write)."""
        if self.on_device:
            return self._dev.ode_table(self._handle, 0, self._states.shape)
        return self._states

    @states.setter
    def states(self, a):
        self._states = np.ascontiguousarray(a, dtype=np.float64).reshape(self._states.shape)
        if self.on_device:
            self._dev.ode_table(self._handle, 0, self._states.shape, upload=self._states)

    @property
    def parameters(self):
        if self.on_device:
            return self._dev.ode_table(self._handle, 1, self._parameters.shape)
        return self._parameters

    @parameters.setter
    def parameters(self, a):
        self._parameters = np.ascontiguousarray(a, dtype=np.float64).reshape(self._parameters.shape)
        if self.on_device:
            self._dev.ode_table(self._handle, 1, self._parameters.shape, upload=self._parameters)

    # --- ODE <- PDE
    def set_state(self, which, u, locator=None):
        return self.__set_ODE('state', which, u, locator=locator)

    def set_parameter(self, which, u, locator=None):
        return self.__set_ODE('parameter', which, u, locator=locator)

    # --- PDE <- ODE
    def get_state(self, which, u, locator=None):
        return self.__get_PDE('state', which, u, locator=locator)

    def get_parameter(self, which, u, locator=None):
        return self.__get_PDE('parameter', which, u, locator=locator)

    def set_state_values(self, value_dict, locator=None):
        return self.__set_ODE_values('state', value_dict, locator=locator)

    def set_parameter_values(self, value_dict, locator=None):
        return self.__set_ODE_values('parameter', value_dict, locator=locator)

    def set_membrane_potential(self, u, locator=None):
        return self.set_state('V', u, locator=locator)

    def get_membrane_potential(self, u, locator=None):
        return self.get_state('V', u, locator=locator)

    @property
    def V_index(self):
        return self.ode.state_indices('V')

    # ---- ODE integration (membrane.py:84-119)
    def step_lsoda(self, dt, stimulus, stimulus_locator=None):
        if stimulus is None:
            stimulus = {}
        if stimulus_locator is None:
            stimulus_locator = _EVERYWHERE
        if self.on_device:
            # The stimulus is re-imposed on the masked rows at the start of every step (membrane.py:98-104) by the
            # device kernel itself; the host only (re)sends mask + values when they change.  The locator is evaluated
            # again whenever a different callable (held by reference, so its identity cannot be recycled) or a different
            # stimulus dict arrives, and the upload is keyed on the mask CONTENTS.
            items = tuple(sorted(stimulus.items()))
            if self._stim_locator is not stimulus_locator or self._stim_mask is None:
                self._stim_mask = np.fromiter(map(stimulus_locator, self.dof_locations), dtype=bool, count=self.nodes)
                self._stim_locator = stimulus_locator
            sig = (items, self._stim_mask.tobytes())
            if sig != self._stim_applied:
                cols = [self.ode.parameter_indices(key) for key, _ in items]
                self._dev.ode_set_stimulus(self._handle, cols, [float(v) for _, v in items], self._stim_mask)
                self._stim_applied = sig
            self._dev.ode_step(self._handle, float(self.time), float(dt), rtol=1.0e-8, atol=0.0)   # membrane.py:112
            self.time = self.time + dt
            return None
        mask = np.fromiter(map(stimulus_locator, self.dof_locations), dtype=bool, count=self.nodes)
        for key, value in stimulus.items():
            self._parameters[mask, self.ode.parameter_indices(key)] = value
        if self.nodes:
            self._states, self._h = integrate_batch(self.ode.rhs, self.time, self.time + dt, self._states,
                                                    self._parameters, rtol=1.0e-8, atol=0.0, h0=self._h)
        self.time = self.time + dt
        return self._states

    # --- work horses (membrane.py:122-186)
    def _lidx(self, locator):
        lidx = np.arange(self.nodes)
        if locator is not None:
            lidx = lidx[np.fromiter(map(locator, self.dof_locations), dtype=bool, count=self.nodes)]
        return lidx

    def _device_field(self, u):
        """(field, row) if `u` lives in a facet field of this model's device, else None."""
        if self.on_device and getattr(u, "dev", None) is self._dev and hasattr(u, "field"):
            if hasattr(u, "check_valid"):
                u.check_valid()
            return u.field, getattr(u, "row", 0)
        return None

    def __set_ODE(self, what, which, u, locator=None):
        get_index = {'state': self.ode.state_indices, 'parameter': self.ode.parameter_indices}[what]
        the_index = get_index(which)
        if self.on_device and locator is None:
            from knpemidg import _abi
            loc = self._device_field(u)
            if loc is None:                                     # host data: the copy must be issued before the staging
                a = u.array() if hasattr(u, "array") else np.asarray(u)     # slot is reused, so do it right away
                self._dev.upload(_abi.F_FACET_TMP, a, 0)
                self._dev.ode_exchange(self._handle, 0 if what == 'state' else 1, the_index, _abi.F_FACET_TMP, 0, to_facet=0)
                self._dev._flush()
                return None
            self._dev.ode_exchange(self._handle, 0 if what == 'state' else 1, the_index, loc[0], loc[1], to_facet=0)
            return None
        destination = self.states if what == 'state' else self.parameters
        lidx = self._lidx(locator)
        source = u.array() if hasattr(u, "array") else np.asarray(u)
        if len(lidx) > 0:
            destination[lidx, the_index] = source[self.indices[lidx]]
        if self.on_device:
            if what == 'state':
                self.states = destination
            else:
                self.parameters = destination
        return destination

    def __get_PDE(self, what, which, u, locator=None):
        get_index = {'state': self.ode.state_indices, 'parameter': self.ode.parameter_indices}[what]
        the_index = get_index(which)
        loc = self._device_field(u)
        if loc is not None and locator is None:
            self._dev.ode_exchange(self._handle, 0 if what == 'state' else 1, the_index, loc[0], loc[1], to_facet=1)
            return u
        source = self.states if what == 'state' else self.parameters
        lidx = self._lidx(locator)
        destination = np.array(u.array(), dtype=np.float64, copy=True)
        if len(lidx) > 0:
            destination[self.indices[lidx]] = source[lidx, the_index]
        u.vector().set_local(destination)
        return u

    def __set_ODE_values(self, what, value_dict, locator=None):
        destination = self.states if what == 'state' else self.parameters
        get_col = self.ode.state_indices if what == 'state' else self.ode.parameter_indices
        lidx = self._lidx(locator)
        if len(lidx) == 0:
            return destination
        coords = self.dof_locations[lidx]
        for param in value_dict:
            col = get_col(param)
            get_value = value_dict[param]
            for row, x in zip(lidx, coords):
                destination[row, col] = get_value(x)
        if self.on_device:
            if what == 'state':
                self.states = destination
            else:
                self.parameters = destination
        return destination
