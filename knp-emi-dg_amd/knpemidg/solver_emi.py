"""`SolverEMI`: the EMI (potential-only) variant -- concentrations are frozen at their initial values, each global
step solves step I and updates the membrane potential (reference: src/knpemidg/solver_emi.py:52-822).  Same device
path as `Solver`; only the sequencing differs."""
import time

from knpemidg.solver import Solver, bcolors


class SolverEMI(Solver):
    def update_ode(self, ode_model):
        # the reference hard-wires the K_e / Na_i traces into its time loop (solver_emi.py:661-668)
        from knpemidg.utils import pcws_constant_project, plus, minus
        K_e = plus(self.c_prev_k.split()[0], self.n_g)
        ode_model.set_parameter('K_e', pcws_constant_project(K_e, self.Q))
        Na_i = minus(self.ion_list[-1]['c'], self.n_g)
        ode_model.set_parameter('Na_i', pcws_constant_project(Na_i, self.Q))

    def solve_for_time_step(self, k, t):
        """Step I only + phi_M = avg JUMP(phi) (solver_emi.py:491-509)."""
        if self.verbose:
            print(f"{bcolors.WARNING} t = {float(t)}  k = {k} {bcolors.ENDC}")
        self.solve_emi()
        # c is untouched, so re-deriving c_elim / E_k from it is the identity; phi_M is the facet average of the new phi
        self.dev.step_updates()
        t.assign(float(t + self.dt))

    def solve_for_time_step_picard(self, k, t):
        raise NotImplementedError("the reference's SolverEMI Picard loop iterates on a KNP solve it does not have")

    def _unpack_solver_params(self, solver_params):
        self.solver_params = solver_params
        self.direct_emi = solver_params.direct_emi
        self.rtol_emi = solver_params.rtol_emi
        self.atol_emi = solver_params.atol_emi
        self.threshold_emi = getattr(solver_params, "threshold_emi", None)
        self.direct_knp = True                       # no KNP system: nothing to precondition
        self.rtol_knp = self.atol_knp = None

    def solve_system_passive(self, Tstop, t, solver_params, membrane_params=None, filename=None):
        self._unpack_solver_params(solver_params)
        self.splitting_scheme = False
        self.filename = filename
        self.save_fields = filename is not None
        self.save_solver_stats = False
        self.setup_varform_emi()
        self.setup_solver_emi()
        self._check_output_args(filename)
        for k in range(int(round(Tstop / float(self.dt)))):
            self.solve_for_time_step(k, t)
            if (k % self.sf) == 0 and self.save_fields:
                self.save_h5()
        if self.save_fields:
            self.close_h5()
        return tuple(self.c.split()) + (self.phi,), self.ion_list[-1]["c"]

    def solve_system_active(self, Tstop, t, solver_params, filename=None):
        self._unpack_solver_params(solver_params)
        self.splitting_scheme = True
        self.filename = filename
        self.save_fields = filename is not None
        self.save_solver_stats = False
        self.setup_varform_emi()
        self.setup_solver_emi()
        self._check_output_args(filename)
        for k in range(int(round(Tstop / float(self.dt)))):
            ts = time.perf_counter()
            self.step_membrane_models(k)
            if self.verbose:
                print(f"{bcolors.OKGREEN} CPU Execution time ODE solve: {time.perf_counter() - ts:.4f} seconds {bcolors.ENDC}")
            self.solve_for_time_step(k, t)
            if (k % self.sf) == 0 and self.save_fields:
                self.save_h5()
        if self.save_fields:
            self.close_h5()
        return
