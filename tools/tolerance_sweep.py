"""Per-step accuracy of the shipped Krylov tolerances against tightly converged solves of the same steps.
usage: tolerance_sweep.py [resolution] [steps] [rtol_emi,...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters
from knpemidg import Constant
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
pairs = [tuple(float(x) for x in v.split("/")) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [(1e-5, 1e-7), (1e-6, 1e-7), (1e-7, 1e-7)]
verbose = os.environ.get("SWEEP_VERBOSE", "0") == "1"
degree = int(os.environ.get("DEGREE", 1))


def run(rtol_emi, rtol_knp):
    S = make_solver(dim=3, resolution=r, verbose=False, degree=degree)
    sp = solver_parameters(3, r)._replace(rtol_emi=rtol_emi, rtol_knp=rtol_knp)
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = Constant(0.0)
    x = S.mesh.coords[S.mesh.cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
    hist = []
    for k in range(steps):
        S.step_membrane_models(k); S.solve_for_time_step(k, t)
        phi = S.phi.array().reshape(S.mesh.num_cells(), -1)
        phi = phi - (phi.mean(axis=1) * vol).sum() / vol.sum()
        hist.append((phi, S.c.array().copy(), S.phi_M_prev_PDE.array().copy()))
    its = (list(S.emi_niter), [max(n) for n in S.knp_niter])
    S.dev.close()
    return hist, its


ref, its_ref = run(1e-11, 1e-13)
print("tight its", its_ref)
for rt, rk in pairs:
    h, its = run(rt, rk)
    print("rtol_emi %.0e rtol_knp %.0e: EMI its %s (sum %d) KNP its %s (sum %d)" % (rt, rk, its[0], sum(its[0]), its[1], sum(its[1])))
    worst = np.zeros(3)
    for k in range(steps):
        (p0, c0, m0), (p1, c1, m1) = h[k], ref[k]
        e = np.array([np.abs(p0 - p1).max() / np.abs(p1).max(), np.abs(c0 - c1).max() / np.abs(c1).max(),
                      np.abs(m0 - m1).max() / np.abs(m1).max()])
        worst = np.maximum(worst, e)
        if verbose:
            print("   step %2d  phi %.2e   c %.2e   phi_M %.2e" % (k, *e))
    print("   worst over %d steps: phi %.2e   c %.2e   phi_M %.2e" % (steps, *worst))
