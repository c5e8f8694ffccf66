"""Per-phase wall time of the first splitting steps on the r=1 idealized mesh (ODE / EMI / KNP / step-III), with a device sync
after every phase: shows the one-off costs of step 0 (code-object loads, graph capture, eigenvalue estimate)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters
from knpemidg import Constant
S = make_solver(dim=3, resolution=1, verbose=False)
S._unpack_solver_params(solver_parameters(3, 1))
S.save_fields = S.save_solver_stats = False
S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
t = Constant(0.0)
for k in range(6):
    S.dev.sync(); t0 = time.perf_counter()
    S.step_membrane_models(k); t1 = time.perf_counter()
    S.solve_emi(); S.dev.sync(); t2 = time.perf_counter()
    S.solve_knp(); S.dev.sync(); t3 = time.perf_counter()
    S.dev.step_updates(); S.dev.sync(); t4 = time.perf_counter()
    print("step %d: ode %.2f emi %.2f knp %.2f upd %.2f ms  its %s %s" % (k, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), S.emi_niter[-1], S.knp_niter[-1]), flush=True)
