"""Iteration counts of the reference's 200-step run at r=2 and where the KNP residual stands at the start and at the end of its solves
(relative to the stopping tolerance): in the quiet phase the extrapolated guess is within 1-2x of the tolerance and the solve runs on to 0.01x
(the early stop under the iteration floor, knp_knp_early_stop).   usage: python tools/knp_floor_histogram.py"""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "examples", "idealized_geometries"))
import numpy as np
from idealized_common import make_solver, solver_parameters, Constant
S = make_solver(dim=3, resolution=2)
S._unpack_solver_params(solver_parameters(3, 2)); S.save_fields = S.save_solver_stats = False; S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
t = Constant(0.0)
res = []
for k in range(200):
    S.step_membrane_models(k); S.solve_for_time_step(k, t)
    r = np.asarray(S.knp_residuals)
    res.append((max(S.knp_niter[-1]), S.emi_niter[-1], float((r[:, 1] / (20 * S._rtol_knp * r[:, 2])).max()), float((r[:, 0] / (20 * S._rtol_knp * r[:, 2])).max())))
its = np.array([x[0] for x in res]); e = np.array([x[1] for x in res])
print("KNP its histogram", {int(v): int((its == v).sum()) for v in np.unique(its)}, "EMI", {int(v): int((e == v).sum()) for v in np.unique(e)})
for k in list(range(40, 200, 8)):
    print(k, "knp its %d emi its %d  final res / tol %.2e  initial res / tol %.2e" % res[k])
