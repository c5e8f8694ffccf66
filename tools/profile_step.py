"""Host-side cProfile of full splitting steps on the r=R idealized 3D mesh (shows where wall time goes outside kernels)."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters
from knpemidg import Constant
r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
S = make_solver(dim=3, resolution=r, verbose=False)
S._unpack_solver_params(solver_parameters(3, r))
S.save_fields = S.save_solver_stats = False
S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
t = Constant(0.0)
S.step_membrane_models(0); S.solve_for_time_step(0, t)
S.dev.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for k in range(1, steps + 1):
    S.step_membrane_models(k); S.solve_for_time_step(k, t)
S.dev.sync()
pr.disable()
print("ms/step %.2f" % ((time.perf_counter() - t0) / steps * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
