"""Host-side cProfile of full splitting steps (shows where wall time goes outside kernels).
usage: profile_step.py <resolution | emix> [steps] [degree]"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters
from knpemidg import Constant
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if len(sys.argv) > 1 and sys.argv[1] == "emix":        # BASELINE configs[4]: the unstructured EMIx reconstruction
    sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
    import emix_common
    S = emix_common.make_solver(degree=degree)
    S._unpack_solver_params(emix_common.solver_parameters())
else:
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S = make_solver(dim=3, resolution=r, verbose=False, degree=degree)
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
    if os.environ.get("KNP_PRINT_RES"):
        print("knp its", S.knp_niter[-1], "res0/|b|", S.knp_residuals[:, 0] / S.knp_residuals[:, 2], "res/|b|", S.knp_residuals[:, 1] / S.knp_residuals[:, 2])
S.dev.sync()
pr.disable()
print("ms/step %.2f  EMI its %s  KNP its %s" % ((time.perf_counter() - t0) / steps * 1e3, S.emi_niter[-steps:], [max(k) for k in S.knp_niter[-steps:]]))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
