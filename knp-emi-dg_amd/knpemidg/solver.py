"""`Solver`: the KNP-EMI splitting solver of adajel/KNP-EMI-DG with its DG assemble-and-solve
hot path running as hand-written HIP kernels on MI355X.

Same class, method names, call order, parameter field names and error behaviour as the
reference (reference: src/knpemidg/solver.py:62-1258); what changed is what sits underneath:

    reference                                     this build
    ---------                                     ----------
    UFL forms + dolfin.assemble (solver.py:477)   matrix-free operator applies  (csrc/apply_p1.hip)
    PETSc Mat build (solver.py:458-460)           nothing -- no matrix exists
    PETSc KSP cg/gmres + hypre (solver.py:509)    device PCG / BiCGStab         (csrc/krylov.hip)
    pcws_constant_project / project (808-845)     facet kernels                 (csrc/rhs_p1.hip)

dolfin objects are replaced by the array stand-ins of `knpemidg.mesh` / `knpemidg.functions`.
There is no CPU fallback: without libknpemi_hip.so and a visible GPU, setup raises.
"""
import os
import sys
import time

import numpy as np

from knpemidg import _abi
from knpemidg.functions import FacetSpace, FacetFunction, DeviceFacetFunction, DeviceFunction
from knpemidg.membrane import MembraneModel
from knpemidg.mesh import Constant
from knpemidg.utils import interface_normal, plus, minus, pcws_constant_project

JUMP = lambda f, n: ("JUMP", f, n)   # symbolic marker; the device evaluates minus - plus


# factor theta of the residual half of the error-controlled EMI stop (Solver._read_solver_params): the solve ends when the TRUE residual,
# in the order-8 norm of its density, is below theta / p^2 * 0.1 rtol_emi * F min|z_k| ||b_knp / vol||_8 (p = polynomial degree: the
# error behind a given residual density grows with the degree, round 2's "DG-P2 needs a 4x tighter potential"; no mesh quantity enters).
# Measured against runs converged to 1e-11 / 1e-13 with BOTH DG-level smoothers of the EMI preconditioner (tools/stop_sweep_r04.py,
# profiles/r04_stop_sweep.txt; bound asked: c <= 1e-6, phi <= 1e-4).
EMI_TARGET_SAFETY = 20.0
# factor on rtol_emi of the energy-norm half of the EMI stop: the solve ends when the estimated ||phi - phi_k||_A (Hestenes-Stiefel
# identity, valid for any SPD preconditioner; csrc/krylov.hip: cg_converged) is below PHI_ENERGY_FACTOR * rtol_emi * ||phi||_A.  It bounds
# the error of the potential itself (north_star: 1e-4 in the max norm at the reference's rtol_emi 1e-5), smooth components included.
PHI_ENERGY_FACTOR = 0.5


class bcolors:
    OKGREEN = '\033[92m'
    WARNING = '\033[93m'
    ENDC = '\033[0m'


def _f(v):
    return float(v)


class Solver:
    def __init__(self, params, ion_list, degree_emi=1, degree_knp=1, mms=None, sf=1):
        self.ion_list = ion_list
        self.N_ions = len(ion_list[:-1])
        self.degree_emi = degree_emi
        self.degree_knp = degree_knp
        self.mms = mms
        self.params = params
        self.sf = sf
        # timers (solver.py:76-81)
        self.ode_solve_timer = 0
        self.emi_solve_timer = 0
        self.knp_solve_timer = 0
        self.emi_ass_timer = 0
        self.knp_ass_timer = 0
        self.verbose = True
        self.dev = None
        self.device_index = int(os.environ.get("LOCAL_RANK", "0"))
        self.mem_models = []
        self.stimulus = None
        self.stimulus_locator = None
        if mms is None and os.environ.get("KNP_NO_AMG", "0") != "1":
            from knpemidg import setup_worker
            setup_worker.prestart(2)       # the two hierarchy helpers import their modules while the caller builds mesh and device context
        self.emi_niter = []
        self.emi_targets = []            # absolute residual targets handed to the EMI solves (error-controlled stop)
        self.knp_niter = []
        # Krylov caps: the reference sets ksp_max_it 1000 behind BoomerAMG (solver.py:429,687);
        # the block-Jacobi preconditioner here needs more, so the cap is a solver_params option.
        self.max_it_emi = 50000
        self.use_amg = os.environ.get("KNP_NO_AMG", "0") != "1"
        self.use_device_ode = os.environ.get("KNP_HOST_ODE", "0") != "1"
        self.max_it_knp = 5000

    # ------------------------------------------------------------------ setup_domain (solver.py:85-121)
    def setup_domain(self, mesh, subdomains, surfaces):
        self.mesh = mesh
        self.subdomains = subdomains
        self.surfaces = surfaces
        self.n_g = interface_normal(subdomains, mesh)
        self.gdim = mesh.geometry().dim()
        self.tau_emi = Constant(20 * self.gdim * self.degree_emi)
        self.tau_knp = Constant(20 * self.gdim * self.degree_knp)
        if self.mms is not None:
            self.lm_tags = [1, 2, 3, 4]
        if self.degree_emi != self.degree_knp:
            raise NotImplementedError("degree_emi must equal degree_knp on the device path")
        return

    # ------------------------------------------------------------------ setup_parameters (solver.py:124-154)
    def setup_parameters(self):
        params = self.params
        self.C_phi = Constant(_f(params.C_phi))
        self.C_M = Constant(_f(params.C_M))
        self.dt = Constant(_f(params.dt))
        self.F = Constant(_f(params.F))
        self.R = Constant(_f(params.R))
        self.temperature = Constant(_f(params.temperature))
        self.psi = _f(self.F) / (_f(self.R) * _f(self.temperature))
        self.phi_M_init_type = params.phi_M_init_type
        for idx, ion in enumerate(self.ion_list):
            ion['D'] = self.make_global(ion['D_sub'])
            self.rho = self.make_global(params.rho_sub)
            if self.mms is not None:
                ion['C'] = self.make_global(ion['C_sub'])
        return

    def make_global(self, f):
        """DG0 array from a {cell tag: value} dict (solver.py:1244-1258); tags absent from the
        dict keep 0 exactly as the reference's zero-initialised Function does."""
        tags = self.subdomains.array()
        q = np.zeros(len(tags), dtype=np.float64)
        for key, value in f.items():
            q[tags == int(key)] = float(value)
        return q

    # ------------------------------------------------------------------ setup_FEM_spaces (solver.py:157-225)
    def setup_FEM_spaces(self):
        d = self.gdim
        p = self.degree_knp
        self.nd = d + 1 if p == 1 else (d + 1) * (d + 2) // 2
        nc = self.mesh.num_cells()
        self._init_c = np.zeros((self.N_ions, nc, self.nd))
        self._init_c_elim = np.zeros((nc, self.nd))
        for idx, ion in enumerate(self.ion_list):
            t = ion['c_init_sub_type']
            if t == 'constant':
                vals = self.make_global(ion['c_init_sub'])[:, None] * np.ones((nc, self.nd))
            elif t == 'expression':
                vals = self.make_global_expression(ion['c_init_sub'])
            elif t == 'function':
                src = ion['c_init_sub']
                vals = np.asarray(src.array() if hasattr(src, "array") else src, dtype=np.float64).reshape(nc, self.nd)
            else:
                print(f"""Type of initial condition \"{t}\" not
                    recognized - please spesify whether initial condition is
                    \"constant\", \"expression\" or \"function\" """)
                sys.exit(0)
            if idx == len(self.ion_list) - 1:
                self._init_c_elim = vals
            else:
                self._init_c[idx] = vals
        self.Q = FacetSpace(self.mesh)
        self._init_phi_M = np.zeros(self.mesh.num_facets())
        if self.phi_M_init_type == 'function':
            src = self.params.phi_M_init
            self._init_phi_M = np.asarray(src.array() if hasattr(src, "array") else src, dtype=np.float64).copy()
        elif self.phi_M_init_type in ('constant', 'expression'):
            pm = getattr(self.params, "phi_M_init", None)
            if pm is not None and self.phi_M_init_type == 'constant':
                # the reference leaves phi_M_prev_PDE = 0 here and takes the value from the ODE file at
                # k == 0 (solver.py:214, 1086-1090); identical once the first ODE step has run.
                pass
        else:
            print(f"""Type of initial condition \"{self.phi_M_init_type}\" not
                recognized - please spesify whether initial condition is
                \"constant\", \"expression\" or \"function\" """)
            sys.exit(0)
        return

    def make_global_expression(self, f):
        """Nodal interpolation of per-subdomain callables f[tag](x) (solver.py:1260-1298)."""
        tags = self.subdomains.array()
        X = self._node_coordinates()
        out = np.zeros(X.shape[:2])
        matched = np.zeros(len(tags), dtype=bool)
        for tag, fun in f.items():
            sel = tags == int(tag)
            matched |= sel
            if sel.any():
                out[sel] = np.asarray(fun(X[sel]), dtype=np.float64)
        assert matched.all(), "Dictionaries for DG data must match cell tags in mesh"
        return out

    def _node_coordinates(self):
        x = self.mesh.coords[self.mesh.cells]
        if self.degree_knp == 1:
            return x
        nv = x.shape[1]
        mids = [0.5 * (x[:, a] + x[:, b]) for a in range(nv) for b in range(a + 1, nv)]
        return np.concatenate([x, np.stack(mids, axis=1)], axis=1)

    # ------------------------------------------------------------------ setup_membrane_model (solver.py:228-267)
    def setup_membrane_model(self, stim_params, odes):
        self.stimulus = stim_params.stimulus
        self.stimulus_locator = stim_params.stimulus_locator
        self.mem_models = []
        self._ensure_device(membrane_tags=[int(t) for t in odes.keys()])
        for tag, ode in odes.items():
            ode_model = MembraneModel(ode, facet_f=self.surfaces, tag=int(tag), V=self.Q)
            ode_model.set_parameter_values({'Cm': lambda x: self.params.C_M})
            on_dev = self.use_device_ode and ode_model.attach_device(self.dev)
            I_ch_k = {}
            for i, ion in enumerate(self.ion_list):
                I_ch_k_ = DeviceFacetFunction(self.Q, self.dev, _abi.F_I_CH, row=i) if on_dev else FacetFunction(self.Q)
                ode_model.get_parameter("I_ch_" + ion['name'], I_ch_k_)
                I_ch_k[ion['name']] = I_ch_k_
            self.mem_models.append({'ode': ode_model, 'I_ch_k': I_ch_k})
        return

    # ------------------------------------------------------------------ device context
    def _ensure_device(self, membrane_tags=None):
        if self.dev is not None:
            return
        if membrane_tags is None:
            membrane_tags = list(self.lm_tags) if self.mms is not None else [m['ode'].tag for m in self.mem_models]
        self.membrane_tags = list(membrane_tags)
        nc = self.mesh.num_cells()
        _abi._stamp("solver: _ensure_device")
        if self.mms is None:
            self._start_knp_helper()
            self._start_emi_helper()
            _abi._stamp("solver: helper jobs handed over")
        self.dev = _abi.Device(self.mesh, self.subdomains.array(), self.surfaces.array(), self.membrane_tags,
                               len(self.ion_list), degree=self.degree_knp, device=self.device_index,
                               nc_owned=getattr(self, "nc_owned", None))
        A = _abi
        dev = self.dev
        self.phi = DeviceFunction(dev, A.F_PHI, nc, self.nd)
        self.c = DeviceFunction(dev, A.F_C, nc, self.nd, n_comp=self.N_ions)
        self.c_prev_k = self.c                                       # identical between Picard levels (solver.py:809)
        self.c_prev_n = DeviceFunction(dev, A.F_C_PREV, nc, self.nd, n_comp=self.N_ions)
        self.ion_list[-1]['c'] = DeviceFunction(dev, A.F_C_ELIM, nc, self.nd)
        self.phi_M_prev_PDE = DeviceFacetFunction(self.Q, dev, A.F_PHI_M)
        for k, ion in enumerate(self.ion_list):
            ion['E'] = DeviceFacetFunction(self.Q, dev, A.F_E, row=k)
        dev.upload(A.F_C, self._init_c)
        dev.copy_field(A.F_C_PREV, A.F_C)                            # the same values: one host-to-device transfer instead of two
        dev.upload(A.F_C_ELIM, self._init_c_elim)
        dev.upload(A.F_PHI_M, self._init_phi_M)
        self._push_params(splitting=True)
        dev.nernst()                                                 # initial E_k (solver.py:299-300)
        _abi._stamp("solver: initial fields and parameters on the device")

    def _push_params(self, splitting):
        z = [float(ion['z']) for ion in self.ion_list]
        D = np.stack([ion['D'] for ion in self.ion_list])
        # ion['f_source'] v dx(0) (solver.py:599): a number / Constant or a per-cell array travels as a DG0 coefficient on the ECS
        # cells; a callable f(x) or f(x, t) (the reference accepts any UFL coefficient, e.g. the box-and-time-window Expression
        # of run_tortuosity.py:180-200) is integrated on the host before every KNP solve (_update_sources)
        fsrc = None
        nc = self.mesh.num_cells()
        ecs = (self.subdomains.array() == 0).astype(np.float64)
        rows, self._callable_sources = [], {}
        for k, ion in enumerate(self.ion_list[:-1]):
            f = ion.get('f_source', 0.0)
            if callable(f) and not hasattr(f, '__float__'):
                self._callable_sources[k] = f
                rows.append(np.zeros(nc))
            elif isinstance(f, np.ndarray) and f.ndim >= 1 and f.size > 1:
                if f.shape != (nc,):
                    raise ValueError("f_source array of ion %s must hold one value per cell" % ion.get('name', k))
                rows.append(f.astype(np.float64) * ecs)
            else:
                rows.append(float(f) * ecs)
        if any(r.any() for r in rows):
            fsrc = np.stack(rows)
        self.dev.set_params(_f(self.C_M), _f(self.dt), _f(self.F), _f(self.R), _f(self.temperature), _f(self.C_phi),
                            _f(self.tau_emi), _f(self.tau_knp), z, D, rho=self.rho, fsrc=fsrc, splitting=splitting)

    # ------------------------------------------------------------------ forms / solvers (solver.py:270-468, 534-721)
    def setup_varform_emi(self):
        """The bilinear / linear forms are fixed kernels; only the splitting flag is data."""
        self._ensure_device()
        if self.mms is not None:
            # manufactured solution (solver.py:349-374, 632-657): data terms integrated once on the host
            from knpemidg.mms_terms import extra_rhs
            Cdev, e_emi, e_knp = extra_rhs(self)
            self.dev.set_mms(Cdev, e_emi, e_knp)
            self._push_params(2)
        else:
            self._push_params(self.splitting_scheme)
        self.I_ch = [None] * len(self.mem_models)
        return

    def setup_varform_knp(self):
        return

    def _update_sources(self, t):
        """Load vector of the callable ion sources at time t: int f_k(x, t) v dx(0) by a degree-8 rule per ECS cell (the reference
        interpolates its Expression to the declared degree and integrates that: run_tortuosity.py:180-200, solver.py:599)."""
        srcs = getattr(self, "_callable_sources", None)
        if not srcs or self.mms is not None:
            return
        import inspect
        from knpemidg.mms_terms import _cell_source, _geometry
        if not hasattr(self, "_src_geom"):
            self._src_geom = (_geometry(self.mesh)[0], np.nonzero(self.subdomains.array() == 0)[0])
        vol, sel = self._src_geom
        out = np.zeros((self.N_ions, self.mesh.num_cells(), self.nd))
        for k, f in srcs.items():
            two = len(inspect.signature(f).parameters) >= 2
            _cell_source(self.mesh, vol, sel, (lambda X, f=f: f(X, float(t))) if two else f, out[k], p=self.degree_knp)
        self.dev.set_source(out)

    def setup_solver_emi(self):
        self._read_solver_params()
        # direct_emi (MUMPS in the reference, solver.py:412-422) is emulated by a tightly converged PCG with the plain
        # block-Jacobi preconditioner: it stays SPD to rounding for any coefficient contrast (the MMS problem couples
        # the membrane with C_phi = 1e10), which a V-cycle does not
        if self.use_amg and not self.direct_emi:
            self.dev.set_emi_dg_smoother(self._emi_dg_chebyshev())
            # the KNP helper usually finishes first (no membrane term, one hierarchy for both species): while the EMI helper is still
            # running, its result is uploaded here instead of in setup_solver_knp (0.2 s of upload hidden behind the wait)
            eh, kh = getattr(self, "_emi_helper", None), getattr(self, "_knp_helper", None)
            if (eh is not None and kh is not None and not self.direct_knp and eh[0]["thread"].is_alive() and not kh[0]["thread"].is_alive()):
                self._setup_amg_knp()
                self._knp_amg_uploaded_early = True
            self._setup_amg_emi()
        else:
            self._drop_emi_helper()
        return

    def _emi_dg_chebyshev(self):
        """DG-level smoother of the EMI preconditioner at setup time: the two-step Chebyshev block-Jacobi (one more operator apply per PCG
        iteration, fewer iterations) unless `emi_dg_chebyshev` in solver_params decides explicitly.  Whether the extra apply pays depends
        on where the run sits between the launch-latency and the bandwidth regime and on the mesh quality (r=2: 4.25 -> 4.7 iterations
        without it for a quarter less work per iteration; EMIx reconstruction: 9.2 -> 13.5 iterations), so it is MEASURED on the first
        EMI system of the run itself (_emi_smoother_trial) instead of being read off mesh-size thresholds (round 3)."""
        sp = getattr(self, "solver_params", None)
        explicit = getattr(sp, "emi_dg_chebyshev", None)
        if explicit is None and os.environ.get("KNP_EMI_CHEB") is not None:
            explicit = int(os.environ["KNP_EMI_CHEB"]) != 0                # the device reads the same variable as an override (csrc/abi.hip)
        self._emi_trial = None if explicit is not None else True
        return True if explicit is None else bool(explicit)

    def _emi_smoother_trial(self, solve):
        """Measured choice of the EMI DG-level smoother, on the FIRST EMI system of the run: the same right-hand side is solved from the
        same initial guess with the Chebyshev step and with plain block-Jacobi (one untimed solve each first: eigenvalue estimate, graph
        capture, code-object loads), each charged its wall time per decade of TRUE-residual reduction; the times are summed over the
        ranks of a partitioned run (one all-reduce: every rank takes the same decision and keeps ONE symmetric preconditioner).  The
        step stays unless dropping it is at least 3 % cheaper.  The time stepping itself never changes preconditioner: a trial spread
        over the first steps of the run (tried first) perturbed the extrapolated initial guesses of the steps behind it (r=2: EMI 4.6 ->
        5.2, KNP 5.1 -> 5.4 iterations per step over the next 20 steps, profiles/r04_smoother_trial.txt).  Both variants meet the same
        stopping test, which does not depend on the preconditioner (csrc/krylov.hip: cg_converged), so the last solve's result IS the
        step's solution.  solve() -> (seconds, niter, res) runs one EMI solve on the current device state."""
        dev = self.dev
        phi0 = dev.download(_abi.F_PHI)
        cost = {}
        out = None
        for timed in (False, True):
            for cheb in (True, False):
                dev.set_emi_dg_smoother(cheb)
                dev.upload(_abi.F_PHI, phi0)                       # same initial guess; a caller-supplied state is no history point
                sec, niter, res = solve()
                if timed:
                    decades = max(np.log10(max(float(res[0]), 1e-300) / max(float(res[1]), 1e-300)), 0.25)
                    cost[cheb] = sec / decades
                out = (sec, niter, res)
        on, off = self.dev.allreduce_sum([cost[True], cost[False]])
        keep = not (off < 0.97 * on)
        self.emi_dg_chebyshev_measured = {"chebyshev_s_per_decade": float(on), "plain_s_per_decade": float(off), "chosen": bool(keep)}
        if self.verbose:
            print(" EMI DG-level smoother: Chebyshev step %.3f ms / decade, plain block-Jacobi %.3f ms / decade -> %s"
                  % (1e3 * on, 1e3 * off, "Chebyshev" if keep else "plain"))
        self._emi_trial = None
        if keep:                                                   # the last solve ran without the step: redo with the chosen one
            dev.set_emi_dg_smoother(True)
            dev.upload(_abi.F_PHI, phi0)
            out = solve()
        return out

    def _host_initial_kappa(self):
        """kappa = F psi sum_k z_k^2 D_k c_k of the initial state, nodal [nc, nd] (what k_kappa computes on the device)."""
        nc = self.mesh.num_cells()
        kappa = np.zeros((nc, self.nd))
        for idx, ion in enumerate(self.ion_list):
            c = self._init_c_elim if idx == len(self.ion_list) - 1 else self._init_c[idx]
            kappa += _f(self.F) * float(ion['z']) ** 2 * self.psi * np.asarray(ion['D'], dtype=np.float64)[:, None] * c
        return kappa

    def _start_emi_helper(self):
        """First EMI hierarchy (from the initial state) in a helper process, next to the KNP one (knpemidg/setup_worker.py).
        Not for partitions (their hierarchy is the global one) and not for manufactured solutions (direct_emi)."""
        if (not self.use_amg or os.environ.get("KNP_AMG_SERIAL_SETUP", "0") == "1" or getattr(self, "_emi_helper", None) is not None
                or getattr(self, "global_mesh_tuple", None) is not None or not hasattr(self, "_init_c")):
            return
        from knpemidg import setup_worker
        kappa = self._host_initial_kappa()
        job = setup_worker.emi_job(self.mesh, self.surfaces.array(), self.membrane_tags, self.degree_knp, kappa, _f(self.params.C_phi))
        self._emi_helper = (setup_worker.start(job), kappa)

    def _drop_emi_helper(self):
        helper = getattr(self, "_emi_helper", None)
        if helper is not None:
            from knpemidg import setup_worker
            self._emi_helper = None
            setup_worker.cancel(helper[0])

    def _setup_amg_emi(self):
        """Preconditioner setup (the reference builds BoomerAMG from BB_emi, solver.py:433, 505): conforming
        P1 operator from the current kappa + smoothed-aggregation hierarchy, built on the host once and reused
        across time steps."""
        from knpemidg import amg
        ts = time.perf_counter()
        dev = self.dev
        levels = dg2cg = None
        helper = getattr(self, "_emi_helper", None)
        if helper is not None:
            # first build: collected from the helper process if the device state still is the initial state it was given
            from knpemidg import setup_worker
            self._emi_helper = None
            handle, kappa_h = helper
            dev.update_kappa()
            kappa = dev.download(_abi.F_KAPPA).reshape(self.mesh.num_cells(), self.nd)
            if np.allclose(kappa, kappa_h, rtol=1e-8, atol=0.0):
                _abi._stamp("solver: waiting for the EMI helper")
                res = setup_worker.collect(handle)
                _abi._stamp("solver: EMI hierarchy collected")
                if res is not None:
                    levels, dg2cg = res["levels"], res["dof"]
                    self._amg_kappa0 = kappa.ravel().copy()
                    self._dg2cg_helper = dg2cg
            else:
                setup_worker.cancel(handle)
        if levels is None:
            gmesh, gsub, gsurf = self._amg_global()
            if gmesh is self.mesh:
                dev.update_kappa()
                kappa = dev.download(_abi.F_KAPPA).reshape(self.mesh.num_cells(), self.nd)
                self._amg_kappa0 = kappa.ravel().copy()          # what the hierarchy was built from (refresh policy)
            else:
                # distributed: every rank builds the SAME global hierarchy from the tag-wise initial state (no
                # communication; the preconditioner is lagged anyway)
                kappa = np.zeros(gmesh.num_cells())
                for ion in self.ion_list:
                    if ion['c_init_sub_type'] != 'constant':
                        raise NotImplementedError("distributed AMG setup needs tag-wise constant initial concentrations")
                    D = self._by_tag(ion['D_sub'], gsub)
                    c0 = self._by_tag(ion['c_init_sub'], gsub)
                    kappa += _f(self.F) * float(ion['z']) ** 2 * self.psi * D * c0
            levels = amg.build_emi_levels(self._cspace, self._cspace2 if self.degree_knp != 1 else None, gsurf.array(),
                                          self.membrane_tags, kappa, _f(self.C_phi))
            dg2cg = self._local_dg2cg()
            d0 = self._dist0()
            if d0 is not None:                      # partitioned run: this rank's rows of the finest conforming level
                local = d0.localize(levels, d0.local_matrix(kappa, membrane_C=_f(self.C_phi)))
                if local is not None:
                    dev.amg_upload(0, d0.local_dg2cg(self.local_mesh.cells_global), local, dist0=True)
                    self.amg_dist0 = getattr(self, "amg_dist0", 0) + 1          # hierarchies uploaded in the row-distributed form
                    levels = None
        if levels is not None:
            dev.amg_upload(0, dg2cg, levels)
            _abi._stamp("solver: EMI hierarchy uploaded")
            nlev = [lv.A.shape[0] for lv in levels]
        else:
            nlev = [lv.A.shape[0] for lv in local]
        self.amg_setup_timer = time.perf_counter() - ts
        if self.verbose:
            print(" AMG(EMI) levels:", nlev, "setup %.2f s" % self.amg_setup_timer)

    def _maybe_refresh_amg_emi(self, niter):
        """Refresh policy of the lagged EMI hierarchy.  The reference rebuilds BoomerAMG from the freshly assembled B_emi at
        EVERY solve (solver.py:479, 505); here the hierarchy is built from kappa at setup time and rebuilt when it has gone
        stale: (a) kappa has moved by more than KNP_AMG_REFRESH_KAPPA (default 25 %) anywhere since the last build (checked
        every KNP_AMG_REFRESH_EVERY = 50 solves), or (b) the PCG iteration count has stayed above 3x its post-build level for 10 consecutive solves.
        (The KNP hierarchies depend on the mesh, D_k and dt only -- nothing to refresh.)  A partitioned run keeps the hierarchy
        of the initial state (every rank would have to gather kappa to rebuild the replicated levels): the Krylov solves still
        converge to their tolerance, only the iteration count can grow."""
        if not (self.use_amg and not self.direct_emi) or getattr(self, "local_mesh", None) is not None:
            return
        st = self.__dict__.setdefault("_amg_refresh", {"solves": 0, "ref": None, "high": 0})
        st["solves"] += 1
        if st["ref"] is None and st["solves"] >= 3:
            st["ref"] = max(1, min(self.emi_niter[-2:]))
        stale = False
        if st["ref"] is not None:
            st["high"] = st["high"] + 1 if niter > 3 * st["ref"] + 2 else 0
            stale = st["high"] >= 10
        if st["solves"] % int(os.environ.get("KNP_AMG_REFRESH_EVERY", 50)) == 0:
            kap, k0 = self.dev.download(_abi.F_KAPPA), self._amg_kappa0
            lim = float(os.environ.get("KNP_AMG_REFRESH_KAPPA", 0.25))
            floor = 1e-12 * max(float(np.abs(k0).max()), 1e-300)                # a zero entry of kappa must not force a rebuild
            drift = np.abs(kap - k0) / np.maximum(np.abs(k0), floor)
            stale = stale or bool(np.nanmax(drift) > lim) or not bool(np.isfinite(kap).all())
        if stale:
            if self.verbose:
                print(" AMG(EMI): hierarchy refreshed (kappa drift / iteration count)")
            self._setup_amg_emi()
            self.amg_refreshes = getattr(self, "amg_refreshes", 0) + 1
            self._amg_refresh = {"solves": 0, "ref": None, "high": 0}

    def setup_solver_knp(self):
        if self.use_amg and not self.direct_knp:
            if self.__dict__.pop("_knp_amg_uploaded_early", False):
                return
            self._setup_amg_knp()
        else:
            self._drop_knp_helper()
        return

    def _drop_knp_helper(self):
        helper = getattr(self, "_knp_helper", None)
        if helper is not None:
            from knpemidg import setup_worker
            self._knp_helper = None
            setup_worker.cancel(helper[0])

    def _amg_global(self):
        """(mesh, subdomains, surfaces) the conforming hierarchy is built on: the global mesh when this solver
        holds one partition of it (make_distributed_solver), else its own mesh."""
        from knpemidg import amg
        g = getattr(self, "global_mesh_tuple", None) or (self.mesh, self.subdomains, self.surfaces)
        if not hasattr(self, "_cspace"):
            self._cspace = amg.ConformingSpace(g[0], g[2].array(), self.membrane_tags)
            if self.degree_knp != 1:
                self._cspace2 = amg.ConformingSpaceP2(self._cspace)
        return g

    def _dist0(self):
        """Row distribution of the finest conforming level over the ranks of a partitioned run (amg.Dist0Space; the reference's BoomerAMG
        is row-distributed by PETSc, solver.py:433, 688), with the shared-dof tables handed to the device on the first call; None on one
        rank and with KNP_AMG_DIST0=0 (replicated level 0 behind an all-reduce of the level-0 residual, rounds 1-2)."""
        loc = getattr(self, "local_mesh", None)
        if loc is None or getattr(loc, "part", None) is None or loc.part.world < 2 or os.environ.get("KNP_AMG_DIST0", "1") == "0":
            return None
        d0 = getattr(self, "_dist0_space", None)
        if d0 is None:
            from knpemidg import amg
            gmesh, gsub, gsurf = self._amg_global()
            space = self._cspace if self.degree_knp == 1 else self._cspace2
            mem = np.nonzero((gmesh.facet_cells[:, 1] >= 0) & np.isin(np.asarray(gsurf.array()), list(self.membrane_tags)))[0]
            d0 = amg.Dist0Space(space, loc.part.owner, mem, loc.rank, loc.part.world)
            self.dev.amg_interface(d0.n, *d0.interface_tables())
            self._dist0_space = d0
        return d0

    def _local_dg2cg(self):
        loc = getattr(self, "local_mesh", None)
        if loc is None and not hasattr(self, "_cspace") and getattr(self, "_dg2cg_helper", None) is not None:
            return self._dg2cg_helper             # spaces built by the helper process only (same deterministic numbering)
        self._amg_global()
        dof = self._cspace.dof if self.degree_knp == 1 else self._cspace2.dof
        return dof if loc is None else dof[loc.cells_global]

    @staticmethod
    def _by_tag(d, subdomains):
        tags = subdomains.array()
        q = np.zeros(len(tags), dtype=np.float64)
        for key, value in d.items():
            q[tags == int(key)] = float(value)
        return q

    def _knp_level0_degree(self):
        # one Jacobi step on the finest conforming level, on one GPU and on partitions alike.  (With a communicator the hierarchy is
        # replicated and the restricted residual is all-reduced; a transfer-only finest level would let that happen on level 1 --
        # 6.5x fewer bytes, no replicated level-0 SpMVs -- but costs 50 % more BiCGStab iterations: 9.9 instead of 6.6 per step in
        # a 4-rank run of the r=2 mesh, which only pays for itself beyond 8 ranks.)
        return 1

    def _start_knp_helper(self):
        """KNP hierarchies depend only on the mesh, D_k and dt: a helper process builds them while this process creates the device
        context and the EMI hierarchy (knpemidg/setup_worker.py: why a process and not a thread).  KNP_AMG_SERIAL_SETUP=1 or a
        failure of the helper: built in this process by setup_solver_knp."""
        if not self.use_amg or os.environ.get("KNP_AMG_SERIAL_SETUP", "0") == "1" or getattr(self, "_knp_helper", None) is not None:
            return
        from knpemidg import setup_worker
        g = getattr(self, "global_mesh_tuple", None) or (self.mesh, self.subdomains, self.surfaces)
        d0 = self._knp_level0_degree()
        job = setup_worker.job_from_solver(g[0], g[1].array(), g[2].array(), self.membrane_tags, self.degree_knp,
                                           [ion['D_sub'] for ion in self.ion_list[:-1]], _f(self.params.dt), d0)
        self._knp_helper = (setup_worker.start(job), d0)

    def _build_amg_knp(self):
        """[(members, levels)] of the KNP preconditioner (amg.build_knp_groups), built in this process."""
        from knpemidg import amg
        gmesh, gsub, gsurf = self._amg_global()
        return amg.build_knp_groups(self._cspace, self._cspace2 if self.degree_knp != 1 else None, gsub.array(),
                                    [ion['D_sub'] for ion in self.ion_list[:-1]], _f(self.dt), self._knp_level0_degree())

    def _setup_amg_knp(self):
        ts = time.perf_counter()
        groups = None
        helper = getattr(self, "_knp_helper", None)
        if helper is not None:
            from knpemidg import setup_worker
            self._knp_helper = None
            handle, d0 = helper
            if d0 == self._knp_level0_degree():
                groups = setup_worker.collect(handle)
            else:                                   # the helper was started with another guess of the communicator state
                setup_worker.cancel(handle)
        if groups is None:
            groups = self._build_amg_knp()
        d0 = self._dist0()
        for members, levels in groups:
            local = None
            if d0 is not None:                      # partitioned run: this rank's rows of the finest conforming level
                gsub = self._amg_global()[1]
                D = np.mean([self._by_tag(self.ion_list[k]['D_sub'], gsub) for k in members], axis=0)
                local = d0.localize(levels, d0.local_matrix(D, mass_coef=np.full(len(D), 1.0 / _f(self.dt))))
            if local is not None:
                self.dev.amg_upload(1 + members[0], d0.local_dg2cg(self.local_mesh.cells_global), local, ncol=len(members), dist0=True)
                self.amg_dist0 = getattr(self, "amg_dist0", 0) + 1
            else:
                self.dev.amg_upload(1 + members[0], self._local_dg2cg(), levels, ncol=len(members))
            for k in members[1:]:
                self.dev.amg_clear(1 + k)
            if self.verbose:
                print(" AMG(KNP %s) levels:" % "+".join(self.ion_list[k]['name'] for k in members), [lv.A.shape[0] for lv in levels])
        self.amg_setup_timer = getattr(self, "amg_setup_timer", 0.0) + time.perf_counter() - ts

    def _read_solver_params(self):
        sp = getattr(self, "solver_params", None)
        self.max_it_emi = int(getattr(sp, "max_it_emi", self.max_it_emi))
        self.max_it_knp = int(getattr(sp, "max_it_knp", self.max_it_knp))
        # direct solvers (MUMPS, solver.py:412-422, 671-681) have no device counterpart: emulate with a tight
        # iterative tolerance
        # PETSc tests ||M^-1 r|| <= rtol ||M^-1 b|| with M = BoomerAMG (solver.py:425-444); with this build's preconditioner the
        # same nominal rtol 1e-5 leaves mean-free phi accurate to 1e-4 of its maximum but the concentrations only to 3e-5
        # (they inherit the RELATIVE error of phi through the drift and membrane terms), measured against direct solves
        # through an action potential (tests/test_gpu_trajectory.py, tools/tolerance_sweep.py).  The nominal tolerance is
        # therefore scaled so that the stated parity bounds (c <= 1e-6, phi <= 1e-4) hold at every step.
        # DG-P2 needs a 4x tighter potential for the same bound on c (configs[2], r=1, 40 steps: worst c 1.8e-6 / 1.1e-6 / 6.2e-7
        # at scale 2e-3 / 1e-3 / 5e-4; a tighter KNP tolerance changes nothing): profiles/r02_tolerance_p2_r1.txt
        # The factor is a property of (mesh family, degree), not a constant: the weaker the preconditioner on a mesh, the larger
        # the error behind a given residual (EMIx reconstruction, 13-36 EMI iterations per solve: worst c 7.1e-6 at the idealized
        # mesh's settings; there the KNP residual test needs tightening too).  `emi_rtol_scale` / `knp_rtol_scale` in
        # solver_params set the factors per configuration (examples/emix_simulations: 1e-4 / 0.03, profiles/r02_tolerance_emix.txt);
        # tools/tolerance_sweep.py / tools/tolerance_emix.py measure them.
        # ROUND 3: the per-mesh factors above are replaced by an error-controlled stop.  The EMI solve stops when its residual, in the
        # cell-volume-weighted norm, is small enough for the concentration accuracy wanted (knp_emi_residual_target,
        # csrc/abi.hip): r_abs = theta * c_tol * F * min_k |z_k| ||b_knp,k||_w with c_tol = 0.1 * rtol_emi (the reference pairs
        # rtol_emi 1e-5 with concentrations good to 1e-6) and ONE safety factor theta for every mesh family and degree
        # (EMI_TARGET_SAFETY below; measured: profiles/r03_error_controlled_stop.txt).  `emi_rtol_scale` in solver_params (or
        # KNP_EMI_RTOL_SCALE, tools only) selects the old preconditioned-norm test with that factor instead.
        dflt = getattr(sp, "emi_rtol_scale", None)
        if dflt is None and "KNP_EMI_RTOL_SCALE" in os.environ:
            dflt = float(os.environ["KNP_EMI_RTOL_SCALE"])
        rt = float(self.rtol_emi) if not self.direct_emi else 0.0
        self._emi_target = None
        if self.direct_emi:
            self._rtol_emi = float(getattr(sp, "rtol_direct", 1e-10))
        elif dflt is not None:
            self._rtol_emi = max(rt * float(dflt), min(rt, 1.0e-11))
        else:
            c_tol = float(getattr(sp, "c_tol", None) or 0.1 * rt)
            theta = float(getattr(sp, "emi_target_safety", None) or os.environ.get("KNP_EMI_TARGET_SAFETY", EMI_TARGET_SAFETY / self.degree_emi ** 2))
            self._emi_target = theta * c_tol
            # Second half of the stop (csrc/krylov.hip: cg_converged): the energy-norm error of the iterate, estimated from the CG
            # coefficients, relative to ||phi||_A.  Round 3 tested the preconditioned norm ||M^-1 r|| here, which under-reports the
            # error by a factor that depends on M (a better preconditioner met it with MORE error left); the Hestenes-Stiefel estimate
            # measures the error itself and holds for any SPD preconditioner.
            self._rtol_emi = float(getattr(sp, "emi_energy_factor", None) or os.environ.get("KNP_EMI_ENERGY_FACTOR", PHI_ENERGY_FACTOR)) * rt
        self._atol_emi = 1e-40 if self.direct_emi else float(self.atol_emi)
        ks = getattr(sp, "knp_rtol_scale", None)
        if ks is None and "KNP_KNP_RTOL_SCALE" in os.environ:
            ks = float(os.environ["KNP_KNP_RTOL_SCALE"])
        kscale = float(ks) if ks is not None else 1.0
        rk = float(self.rtol_knp) if not self.direct_knp else 0.0
        # direct_knp (MUMPS in the reference) is emulated by a tightly converged solve: rtol_direct is meant for the residual itself, so the
        # factor 20 the device puts on the order-8 density test (csrc/abi.hip: KNP_D8_FACTOR) is divided out here
        self._rtol_knp = float(getattr(sp, "rtol_direct", 1e-10)) / 20.0 if self.direct_knp else max(rk * kscale, min(rk, 1.0e-13))
        if self.verbose:
            print(" effective tolerances: EMI %s, KNP rtol %.2e (weighted residual norm)" % (
                ("residual target %.2e x F min|z| ||b_knp||" % self._emi_target) if self._emi_target else "rtol %.2e" % self._rtol_emi,
                self._rtol_knp))
        self._atol_knp = 1e-40 if self.direct_knp else float(self.atol_knp)
        # KNP Krylov method: the device default is BiCGStab; `knp_krylov = "gmres"` (+ `gmres_restart`, default 30) in solver_params,
        # or KNP_KNP_KRYLOV=gmres, selects the reference's restarted GMRES (ksp_type gmres / ksp_gmres_restart 30, solver.py:684-701)
        meth = getattr(sp, "knp_krylov", None) or os.environ.get("KNP_KNP_KRYLOV", "bicgstab")
        # ksp_min_it 5 of the reference's GMRES (solver.py:686) = at least five operator + preconditioner applications.  A BiCGStab
        # iteration makes two of each, so three iterations (six applications) would honour it; rounds 1-3 forced five iterations = ten
        # applications, and the solves of the quiet phases of a run stopped at exactly that floor.  FOUR is the measured optimum: the
        # forced iterations of the quiet steps are what keeps the per-step errors from adding up over a run (r=1, 40 steps, worst c
        # against tight solves: min_it 3: 9.0e-7 at 3.95 iterations per step, 4: 2.2e-7 at 4.47, 5: 2.1e-7 at 5.12; tightening the
        # residual test instead buys less per iteration: factor 10 at min_it 3: 5.8e-7 at 4.70; profiles/r04_min_it.txt,
        # r04_stop_sweep.txt).  KNP_KNP_MIN_IT / solver_params.knp_min_it change it.
        self._knp_min_it = int(getattr(sp, "knp_min_it", None) or os.environ.get("KNP_KNP_MIN_IT", 5 if meth == "gmres" else 4))
        # The floor is there for the quiet phases; a solve whose residual is already 100x UNDER the tolerance needs no forced iterations
        # for that purpose (knp_knp_early_stop).  Over the reference's 200-step run at r=2: 4.20 -> 3.54 KNP iterations per step, KNP time
        # -16 %, worst c / phi over 100 steps unchanged to three digits at r=1 and r=2 (2.31e-7 / 2.15e-7: they come from the action
        # potential, where the floor does not bind); a factor 0.1 instead of 0.01: 2.57 iterations but c 7.2e-7 at r=1
        # (profiles/r04_early_stop_sweep.txt).  solver_params.knp_early_stop / KNP_KNP_EARLY set the factor, 0 switches it off.
        early = getattr(sp, "knp_early_stop", None)
        if early is None:
            early = float(os.environ.get("KNP_KNP_EARLY", 0.01))
        self._knp_early = 0.0 if self.direct_knp else float(early)
        if self.dev is not None:
            self.dev.knp_early_stop(self._knp_early)
        if self.dev is not None and meth != getattr(self, "_knp_krylov", "bicgstab"):
            self.dev.set_knp_krylov(meth, int(getattr(sp, "gmres_restart", None) or os.environ.get("KNP_GMRES_RESTART", 30)))
            self._knp_krylov = meth

    def _sync_membrane_to_device(self):
        """phi_M and I_ch_k facet fields produced by the ODE step -> device."""
        if not self.mem_models or all(mm['ode'].on_device for mm in self.mem_models):
            return
        A = _abi
        nf = self.mesh.num_facets()
        Ich = np.zeros((len(self.ion_list), nf))
        if any(mm['ode'].on_device for mm in self.mem_models):
            Ich = self.dev.download(A.F_I_CH).reshape(len(self.ion_list), nf)
        for mm in self.mem_models:
            if mm['ode'].on_device:
                continue
            idx = mm['ode'].indices
            for k, ion in enumerate(self.ion_list):
                Ich[k, idx] = mm['I_ch_k'][ion['name']].array()[idx]
        self.dev.upload(A.F_I_CH, Ich)

    # ------------------------------------------------------------------ solve_emi (solver.py:470-531)
    def solve_emi(self):
        dev = self.dev
        ts = time.perf_counter()
        dev.update_kappa()
        dev.emi_rhs()
        dev.sync()
        te = time.perf_counter()
        res = te - ts
        if self.verbose:
            print(f"{bcolors.OKGREEN} GPU Execution time PDE assemble emi: {res:.4f} seconds {bcolors.ENDC}")
        self.emi_ass_timer += res
        if self.save_solver_stats:
            self.file_emi_assem.write("ass_time: %.4f \n" % (res))
        ts = time.perf_counter()
        if self._emi_target:
            r_abs = self._emi_target * _f(self.F) * self._knp_load_norm()
            self.emi_targets.append(r_abs)            # identical on every rank of a partitioned run (tests/test_gpu_multirank.py)
            dev.emi_residual_target(r_abs)
        if getattr(self, "_emi_trial", None) is not None and self._emi_target and self.use_amg and not self.direct_emi:
            def one_solve():
                t0 = time.perf_counter()
                n_, r_ = dev.emi_solve(self._rtol_emi, self._atol_emi, maxit=self.max_it_emi)
                return time.perf_counter() - t0, n_, r_
            _, niter, r = self._emi_smoother_trial(one_solve)
        else:
            niter, r = dev.emi_solve(self._rtol_emi, self._atol_emi, maxit=self.max_it_emi)
        te = time.perf_counter()
        res = te - ts
        if self.verbose:
            print(f"{bcolors.OKGREEN} GPU Execution time PDE solve emi: {res:.4f} seconds ({niter} its) {bcolors.ENDC}")
        self.emi_solve_timer += res
        self.emi_niter.append(niter)
        self._maybe_refresh_amg_emi(niter)
        if self.save_solver_stats:
            if not self.direct_emi:
                self.file_emi_niter.write("niter: %d \n" % niter)
            self.file_emi_solve.write("solve_time: %.4f \n" % (res))
        return

    def _knp_load_norm(self):
        """min_k |z_k| ||b_knp,k / vol||_8: size of the KNP load vectors (~ M c_k / dt) in the density norm of order 8 the solvers test
        (csrc/krylov.hip) -- from the previous KNP solve; before the first one, from the right-hand side of the initial state."""
        z = np.abs([float(ion['z']) for ion in self.ion_list[:-1]])
        bn = getattr(self, "_knp_bnorm", None)
        if bn is None:
            dev = self.dev
            _abi._stamp("first step: EMI right-hand side ready")
            dev.update_dnphi(); dev.knp_rhs()
            # summed over the owned cells on the device (a partition all-reduces the sums: every rank must hand the SAME target to the PCG
            # stopping test -- ranks that disagree about convergence leave the loop of collectives at different iterations); a host
            # pass over the downloaded field cost 2 s at 8 x 10^6 cells
            sums = dev.allreduce_sum(dev.knp_load_measure())
            bn = np.sqrt(sums) if os.environ.get("KNP_KNP_NORM2", "0") == "1" else sums ** 0.125      # order-8 norm of the load density
            self._knp_bnorm = bn
            _abi._stamp("first step: KNP load norm of the initial state")
        return float(np.min(z * np.asarray(bn)))

    # ------------------------------------------------------------------ solve_knp (solver.py:723-791)
    def solve_knp(self):
        dev = self.dev
        ts = time.perf_counter()
        self._update_sources(getattr(self, "_t_now", 0.0))
        dev.update_dnphi()
        dev.knp_rhs()
        dev.sync()
        te = time.perf_counter()
        res = te - ts
        if self.verbose:
            print(f"{bcolors.OKGREEN} GPU Execution time PDE assemble knp: {res:.4f} seconds {bcolors.ENDC}")
        self.knp_ass_timer += res
        if self.save_solver_stats:
            self.file_knp_assem.write("ass_time: %.4f \n" % (res))
        ts = time.perf_counter()
        niters, r = dev.knp_solve(self._rtol_knp, self._atol_knp, maxit=self.max_it_knp, min_it=self._knp_min_it)
        self._knp_bnorm = np.asarray(r)[:, 2].copy()
        te = time.perf_counter()
        res = te - ts
        if self.verbose:
            print(f"{bcolors.OKGREEN} GPU Execution time PDE solve knp: {res:.4f} seconds ({niters} its) {bcolors.ENDC}")
        self.knp_solve_timer += res
        self.knp_niter.append(niters)
        self.knp_residuals = r                       # per species: initial residual, final residual, |b|
        if self.save_solver_stats:
            self.file_knp_solve.write("solve_time: %.4f \n" % (res))
            if not self.direct_knp:
                self.file_knp_niter.write("niter: %d \n" % max(niters))
        return

    # ------------------------------------------------------------------ solve_for_time_step (solver.py:794-847)
    def solve_for_time_step(self, k, t):
        if self.verbose:
            print("------------------------------------------------")
            print(f"{bcolors.WARNING} t = {float(t)} {bcolors.ENDC}")
            print(f"{bcolors.WARNING} k = {k} {bcolors.ENDC}")
            print("------------------------------------------------")
        if self.mms is not None and getattr(self.mms, "time_dependent", False):
            from knpemidg.mms_terms import extra_rhs       # data terms at the current t (solver.py:845 advances t last)
            self.dev.set_mms(*extra_rhs(self))
        self._t_now = float(t)              # time-dependent sources see the t of this step (solver.py:845 advances t last)
        first = k == 0 and os.environ.get("KNP_DEBUG_SETUP", "0") == "1"       # what the very first step spends on one-time work
        if first:
            self.dev.sync(); _abi._stamp("first step: start")
        self.solve_emi()                    # step I
        if first:
            self.dev.sync(); _abi._stamp("first step: EMI solve (block-Jacobi build, smoother trial, graph capture)")
        self.solve_knp()                    # step II
        if first:
            self.dev.sync(); _abi._stamp("first step: KNP solve")
        self.dev.step_updates()             # step III: c_prev <- c, phi_M, E_k, c_elim
        if first:
            self.dev.sync(); _abi._stamp("first step: step III")
        t.assign(float(t + self.dt))
        return

    # ------------------------------------------------------------------ Picard variant (solver.py:850-927)
    def solve_for_time_step_picard(self, k, t):
        """One global step with inner Picard iterations on the two PDE solves (tolerance 1e-4 on the inf-norm of the
        concentration update, at most 25 iterations, then sys.exit(2) -- reference semantics)."""
        if self.verbose:
            print(f"{bcolors.WARNING} t = {float(t)}  k = {k} (Picard) {bcolors.ENDC}")
        t.assign(float(t + self.dt))
        self._t_now = float(t)
        tol, eps, max_iter, it = 1.0e-4, 2.0, 25, 0
        A = _abi
        while eps > tol:
            it += 1
            self.dev.copy_field(A.F_X, A.F_C)               # c_prev_k of this Picard level
            self.solve_emi()
            self.solve_knp()
            eps = self.dev.max_abs_diff(A.F_X, A.F_C)
            self.dev.picard_updates()                       # c_prev_k <- c (same field), E_k, c_elim for the next level
            if it > max_iter:
                print("Picard solver diverged")
                sys.exit(2)
        self.dev.step_updates()                             # c_prev_n <- c_prev_k, phi_M, (E_k, c_elim unchanged)
        self.picard_iters = getattr(self, "picard_iters", []) + [it]
        if self.verbose:
            print(f" Summary Picard: eps = {eps}, #iters = {it}")
        return

    def _unpack_solver_params(self, solver_params):
        self.solver_params = solver_params
        self.direct_emi = solver_params.direct_emi
        if not self.direct_emi:
            self.rtol_emi = solver_params.rtol_emi
            self.atol_emi = solver_params.atol_emi
            self.threshold_emi = solver_params.threshold_emi      # accepted and ignored (no hypre)
        self.direct_knp = solver_params.direct_knp
        if not self.direct_knp:
            self.rtol_knp = solver_params.rtol_knp
            self.atol_knp = solver_params.atol_knp
            self.threshold_knp = solver_params.threshold_knp

    def _check_output_args(self, filename):
        if filename is None and (self.save_solver_stats or self.save_fields):
            print("Please specify filename when initiating Solver.solve_system_*() method")
            sys.exit(0)
        if self.save_fields:
            self.init_h5_savefile(filename + 'results.h5')                          # solver.py:978, 1063
        if self.save_solver_stats:
            self.init_solver_stats(filename + 'solver/')

    # ------------------------------------------------------------------ solve_system_passive (solver.py:930-1011)
    def solve_system_passive(self, Tstop, t, solver_params, membrane_params, filename=None, save_fields=False,
                             save_solver_stats=False):
        self.filename = filename
        self.save_fields = save_fields
        self.save_solver_stats = save_solver_stats
        self._unpack_solver_params(solver_params)
        self.splitting_scheme = False
        self.setup_varform_emi()
        self.setup_varform_knp()
        self.setup_solver_emi()
        self.setup_solver_knp()
        self._check_output_args(filename)
        for k in range(int(round(Tstop / float(self.dt)))):
            self.solve_for_time_step(k, t)
            if (k % self.sf) == 0 and self.save_fields:
                self.save_h5()
        if self.save_fields:
            self.close_h5()
        if self.save_solver_stats:
            self.close_solver_stats()
        uh = self.c.split() + (self.phi,)
        return uh, self.ion_list[-1]['c']

    # ------------------------------------------------------------------ solve_system_active (solver.py:1014-1135)
    def solve_system_active(self, Tstop, t, solver_params, filename=None, save_fields=False, save_solver_stats=False):
        self.filename = filename
        self.save_fields = save_fields
        self.save_solver_stats = save_solver_stats
        self._unpack_solver_params(solver_params)
        self.splitting_scheme = True
        self.setup_varform_emi()
        self.setup_varform_knp()
        self.setup_solver_emi()
        self.setup_solver_knp()
        self._check_output_args(filename)
        for k in range(int(round(Tstop / float(self.dt)))):
            self.step_membrane_models(k)
            self.solve_for_time_step(k, t)
            if (k % self.sf) == 0 and self.save_fields:
                self.save_h5()
        if self.save_fields:
            self.close_h5()
        if self.save_solver_stats:
            self.close_solver_stats()
        return

    def step_membrane_models(self, k):
        """ODE step of every membrane model + PDE<->ODE copies (solver.py:1076-1118)."""
        ts = time.perf_counter()
        dt_ode = float(self.dt)
        for mem_model in self.mem_models:
            ode_model = mem_model['ode']
            if (self.phi_M_init_type == 'constant') and (k == 0):
                pass
            else:
                ode_model.set_membrane_potential(self.phi_M_prev_PDE)
            for i, ion in enumerate(self.ion_list):
                ode_model.set_parameter(f"E_{ion['name']}", ion['E'])
            self.update_ode(ode_model)
            ode_model.step_lsoda(dt=dt_ode, stimulus=self.stimulus, stimulus_locator=self.stimulus_locator)
            ode_model.get_membrane_potential(self.phi_M_prev_PDE)
            for ion, I_ch_k in mem_model['I_ch_k'].items():
                ode_model.get_parameter("I_ch_" + ion, I_ch_k)
        self._sync_membrane_to_device()
        res = time.perf_counter() - ts
        self.ode_solve_timer += res
        if self.verbose:
            print(f"{bcolors.OKGREEN} CPU Execution time ODE solve: {res:.4f} seconds {bcolors.ENDC}")

    def update_ode(self, ode_model):
        raise NotImplementedError("Subclasses must implement the 'update_ode' function.")

    # ------------------------------------------------------------------ solver statistics (solver.py:1146-1211)
    def init_solver_stats(self, path_timings):
        os.makedirs(path_timings, exist_ok=True)
        reso = getattr(self.solver_params, "resolution", 0)
        num_cells = self.mesh.num_cells()
        dofs_emi = num_cells * self.nd
        dofs_knp = dofs_emi * self.N_ions
        sfx_e = "_dir" if self.direct_emi else ""
        sfx_k = "_dir" if self.direct_knp else ""
        self.file_emi_solve = open(path_timings + "emi_solve%s_%d.txt" % (sfx_e, reso), "w")
        self.file_emi_assem = open(path_timings + "emi_assem%s_%d.txt" % (sfx_e, reso), "w")
        self.file_emi_niter = None if self.direct_emi else open(path_timings + "emi_niter_%d.txt" % reso, "w")
        self.file_knp_solve = open(path_timings + "knp_solve%s_%d.txt" % (sfx_k, reso), "w")
        self.file_knp_assem = open(path_timings + "knp_assem%s_%d.txt" % (sfx_k, reso), "w")
        self.file_knp_niter = None if self.direct_knp else open(path_timings + "knp_niter_%d.txt" % reso, "w")
        for f, dofs in ((self.file_emi_solve, dofs_emi), (self.file_emi_assem, dofs_emi), (self.file_emi_niter, dofs_emi),
                        (self.file_knp_solve, dofs_knp), (self.file_knp_assem, dofs_knp), (self.file_knp_niter, dofs_knp)):
            if f is not None:
                f.write("num cells: %d \n" % num_cells)
                f.write("dofs: %d \n" % dofs)

    def close_solver_stats(self):
        for f in (self.file_emi_niter, self.file_knp_niter, self.file_emi_solve, self.file_knp_solve,
                  self.file_emi_assem, self.file_knp_assem):
            if f is not None:
                f.close()

    # ------------------------------------------------------------------ field output (solver.py:1214-1242)
    # One HDF5 file `<filename>.h5` with the reference's dataset names (written by knpemidg.h5lite.H5Writer; no h5py here):
    #   /mesh/coordinates, /mesh/topology, /subdomains/values, /surfaces/values (+ /surfaces/topology: facet vertices),
    #   /concentrations/vector_n, /elim_concentration/vector_n, /potential/vector_n   (n = 0: initial state, then every sf-th step)
    # plus this build's DG dof layout per field (cell_dofs / x_cell_dofs as DOLFIN stores them): dof(c, j) = c nd + j, species-major
    # for the mixed concentration function.  A distributed run writes one file per rank (`..._rank<r>.h5`) holding its owned +
    # ghost cells and /cells_global, the local -> global cell map.
    def init_h5_savefile(self, filename):
        from knpemidg.h5lite import H5Writer
        self.h5_idx = 0
        os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
        loc = getattr(self, "local_mesh", None)
        rank_sfx = "" if loc is None else "_rank%d" % getattr(self.dev, "rank", loc.rank)
        stem, ext = os.path.splitext(filename)                      # the reference passes the full file name (solver.py:978, 1214)
        self.h5_path = stem + rank_sfx + (ext or ".h5")
        w = self.h5_file = H5Writer(self.h5_path)
        w.write("/mesh/coordinates", self.mesh.coords)
        w.write("/mesh/topology", self.mesh.cells.astype(np.int64))
        w.write("/subdomains/values", np.asarray(self.subdomains.array()).astype(np.uint64))
        w.write("/surfaces/values", np.asarray(self.surfaces.array()).astype(np.uint64))
        w.write("/surfaces/topology", self.mesh.facets.astype(np.int64))
        if loc is not None:
            w.write("/cells_global", loc.cells_global.astype(np.int64))
        nc, nd = self.mesh.num_cells(), self.nd
        base = (np.arange(nc)[:, None] * nd + np.arange(nd)[None, :]).astype(np.int64)
        for name, ncomp in (("concentrations", self.N_ions), ("elim_concentration", 1), ("potential", 1)):
            dofs = np.concatenate([base + k * nc * nd for k in range(ncomp)], axis=1)
            w.write("/%s/cell_dofs" % name, dofs.ravel())
            w.write("/%s/x_cell_dofs" % name, (np.arange(nc + 1) * dofs.shape[1]).astype(np.int64))
            w.write("/%s/cells" % name, np.arange(nc, dtype=np.int64))
        self._write_snapshot()

    def _write_snapshot(self):
        w = self.h5_file
        w.write("/concentrations/vector_%d" % self.h5_idx, self.c.array().ravel())
        w.write("/elim_concentration/vector_%d" % self.h5_idx, self.ion_list[-1]['c'].array().ravel())
        w.write("/potential/vector_%d" % self.h5_idx, self.phi.array().ravel())

    def save_h5(self):
        self.h5_idx += 1
        self._write_snapshot()

    def close_h5(self):
        self.h5_file.close()
        return
