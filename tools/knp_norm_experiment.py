"""Which residual norm tracks the max-norm error of the concentrations?  (round 3, behind the KNP stopping test)
For one KNP system in the middle of a stimulated run (idealized r=1 P1 and the EMIx reconstruction): BiCGStab stopped after k = 1, 2, ...
iterations from the same initial guess; for every k the true error against the converged solution (max norm relative to the species'
maximum, the metric of the parity tests) next to candidate residual measures computed from r = b - A c on the host:
  w2   ||r||_w / ||b||_w                     (cell-volume-weighted 2-norm)
  m8   power mean (order 8) of ||r_K|| / ||b_K||
  d8   ||r / vol||_8 / ||b / vol||_8         (order-8 norms of the residual / load DENSITIES)
  dmax max_K ||r_K|| / vol_K  /  max_K ||b_K|| / vol_K
usage: knp_norm_experiment.py [P1|emix] [n_steps_before]"""
import os, sys
os.environ["KNP_EXTRAPOLATE"] = "0"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "emix_simulations"), os.path.join(ROOT, "examples", "idealized_geometries")]
from knpemidg import _abi as A
from knpemidg._abi import KnpError
cfg = sys.argv[1] if len(sys.argv) > 1 else "P1"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
if cfg == "emix":
    import emix_common as E
    S = E.make_solver(); sp = E.solver_parameters()._replace(rtol_emi=1e-9, rtol_knp=1e-11); t = E.Constant(0.0)
else:
    import idealized_common as I
    S = I.make_solver(dim=3, resolution=1, n_axons=4); sp = I.solver_parameters(3, 1)._replace(rtol_emi=1e-9, rtol_knp=1e-11); t = I.Constant(0.0)
S._unpack_solver_params(sp); S.verbose = False
S.save_fields = S.save_solver_stats = False; S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
for k in range(nsteps):
    S.step_membrane_models(k); S.solve_for_time_step(k, t)
dev = S.dev
S.step_membrane_models(nsteps)
S.solve_emi()
dev.update_dnphi(); dev.knp_rhs()
nc, nd, ns = S.mesh.num_cells(), S.nd, S.N_ions
b = dev.download(A.F_B_KNP).reshape(ns, nc, nd)
c0 = dev.download(A.F_C).copy()
x = S.mesh.coords[S.mesh.cells]; vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
dev.knp_solve(1e-14, maxit=400)
cref = dev.download(A.F_C).reshape(ns, nc, nd).copy()
bK = np.sqrt((b ** 2).sum(axis=2))
print(cfg, "cells", nc, "vol range %.1e .. %.1e" % (vol.min(), vol.max()))
print(" k   err(max-norm)   w2        m8        d8        dmax")
for k in list(range(1, 16)) + [18, 22, 26, 30]:
    dev.upload(A.F_C, c0)
    try:
        dev.knp_solve(1e-30, maxit=k, min_it=0)
    except KnpError:
        pass
    ck = dev.download(A.F_C).reshape(ns, nc, nd)
    dev.upload(A.F_X, ck); dev.knp_apply(A.F_X, A.F_Y)
    r = b - dev.download(A.F_Y).reshape(ns, nc, nd)
    rK = np.sqrt((r ** 2).sum(axis=2))
    err = max(np.abs(ck[s] - cref[s]).max() / np.abs(cref[s]).max() for s in range(ns))
    w2 = max(np.sqrt((rK[s] ** 2 / vol).sum() / (bK[s] ** 2 / vol).sum()) for s in range(ns))
    m8 = max(np.mean((rK[s] / bK[s]) ** 8) ** 0.125 for s in range(ns))
    d8 = max((np.mean((rK[s] / vol) ** 8) / np.mean((bK[s] / vol) ** 8)) ** 0.125 for s in range(ns))
    dm = max((rK[s] / vol).max() / (bK[s] / vol).max() for s in range(ns))
    print("%2d   %.2e       %.2e  %.2e  %.2e  %.2e" % (k, err, w2, m8, d8, dm), flush=True)
