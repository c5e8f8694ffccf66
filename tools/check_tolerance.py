"""Accuracy of the default (reference) Krylov tolerances: the same steps solved at rtol 1e-5 / 1e-7 and at 1e-11 / 1e-12.
usage: check_tolerance.py [resolution] [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver, solver_parameters
from knpemidg import Constant
r = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = []
for tight in (False, True):
    S = make_solver(dim=3, resolution=r, verbose=False)
    sp = solver_parameters(3, r)
    if tight:
        sp = sp._replace(rtol_emi=1e-11, rtol_knp=1e-12)
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = Constant(0.0)
    for k in range(steps):
        S.step_membrane_models(k); S.solve_for_time_step(k, t)
    phi = S.phi.array().reshape(S.mesh.num_cells(), -1)
    x = S.mesh.coords[S.mesh.cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
    phi = phi - (phi.mean(axis=1) * vol).sum() / vol.sum()
    out.append((phi, S.c.array().copy(), S.phi_M_prev_PDE.array().copy(), list(S.emi_niter), [max(n) for n in S.knp_niter]))
    S.dev.close()
(p0, c0, m0, e0, k0), (p1, c1, m1, e1, k1) = out
print("EMI its default", e0, "tight", e1)
print("KNP its default", k0, "tight", k1)
print("rel. difference phi %.2e   c %.2e   phi_M %.2e" % (np.abs(p0 - p1).max() / np.abs(p1).max(), np.abs(c0 - c1).max() / np.abs(c1).max(),
                                                      np.abs(m0 - m1).max() / np.abs(m1).max()))
