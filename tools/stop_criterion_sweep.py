"""Calibration / evidence of the error-controlled EMI stop (round 3): for the three configuration families -- idealized P1 (r=1),
idealized P2 (r=1 = configs[2]), the EMIx reconstruction (configs[4]) -- the worst relative max-norm errors of phi (mean-free), c and phi_M
over a stimulated run against the same run converged to rtol 1e-11 / 1e-13, for a list of safety factors theta of the residual
target and, for comparison, for the round-2 per-mesh factors on the preconditioned norm.
usage: stop_criterion_sweep.py [theta ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "emix_simulations"), os.path.join(ROOT, "examples", "idealized_geometries")]
import emix_common as E
import idealized_common as I

thetas = [float(a) for a in sys.argv[1:]] or [30.0, 10.0, 3.0, 1.0]


def build(cfg, extra):
    if cfg == "emix":
        S = E.make_solver(); sp = E.solver_parameters()
    else:
        S = I.make_solver(dim=3, resolution=1, n_axons=4, degree=2 if cfg == "P2" else 1)
        sp = I.solver_parameters(3, 1)
    if extra:
        fields = sp._asdict(); fields.update(extra)
        from collections import namedtuple
        sp = namedtuple("solver_params", fields.keys())(*fields.values())
    S._unpack_solver_params(sp)
    S.verbose = False
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    return S, (E.Constant if cfg == "emix" else I.Constant)(0.0)


def run(cfg, steps, extra):
    S, t = build(cfg, extra)
    x = S.mesh.coords[S.mesh.cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
    hist = []
    for k in range(steps):
        S.step_membrane_models(k); S.solve_for_time_step(k, t)
        phi = S.phi.array().reshape(S.mesh.num_cells(), -1)
        phi = phi - (phi.mean(axis=1) * vol).sum() / vol.sum()
        hist.append((phi, S.c.array().copy(), S.phi_M_prev_PDE.array().copy()))
    its = (np.mean(S.emi_niter), np.mean([max(n) for n in S.knp_niter]))
    S.dev.close()
    return hist, its


def worst(h, ref):
    w = np.zeros(3)
    for (p0, c0, m0), (p1, c1, m1) in zip(h, ref):
        mem = np.nonzero(m1)[0]
        w = np.maximum(w, [np.abs(p0 - p1).max() / np.abs(p1).max(), np.abs(c0 - c1).max() / np.abs(c1).max(),
                           np.abs(m0[mem] - m1[mem]).max() / np.abs(m1[mem]).max()])
    return w


OLD = {"P1": dict(emi_rtol_scale=2e-3), "P2": dict(emi_rtol_scale=5e-4), "emix": dict(emi_rtol_scale=1e-4, knp_rtol_scale=0.03)}
for cfg, steps in (("P1", 40), ("P2", 25), ("emix", 25)):
    ref, its = run(cfg, steps, dict(rtol_emi=1e-11, rtol_knp=1e-13))
    print("%s: reference run EMI %.1f / KNP %.1f its per step" % (cfg, *its), flush=True)
    h, its = run(cfg, steps, OLD[cfg])
    print("  round-2 factors %s: EMI %.1f KNP %.1f its | worst phi %.2e c %.2e phi_M %.2e" % (OLD[cfg], *its, *worst(h, ref)), flush=True)
    for th in thetas:
        h, its = run(cfg, steps, dict(emi_target_safety=th))
        print("  theta %.3g: EMI %.1f KNP %.1f its | worst phi %.2e c %.2e phi_M %.2e" % (th, *its, *worst(h, ref)), flush=True)
