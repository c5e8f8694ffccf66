"""Calibration / evidence of the round-4 EMI stop (true-residual target theta + energy-norm error estimate g): for one configuration
family -- idealized P1 (r=1), idealized P2 (r=1 = configs[2]), the EMIx reconstruction (configs[4]), the 2D neuron (configs[0]) -- and
one DG-level smoother of the EMI preconditioner (emi_dg_chebyshev 0 / 1), the worst relative max-norm errors of phi (mean-free), c and
phi_M over a stimulated run against the same run converged to rtol 1e-11 / 1e-13, for a list of (theta, g) pairs.
usage: stop_sweep_r04.py CFG CHEB STEPS theta/g [theta/g ...]        CFG in P1 P2 emix 2D; environment switches apply to every run"""
import os, sys
from collections import namedtuple
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "emix_simulations"), os.path.join(ROOT, "examples", "idealized_geometries")]
import emix_common as E
import idealized_common as I

cfg, cheb, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
pairs = [tuple(float(v) for v in a.split("/")) for a in sys.argv[4:]]


def build(extra):
    if cfg == "emix":
        S = E.make_solver(); sp = E.solver_parameters()
    elif cfg == "2D":
        S = I.make_solver(dim=2, resolution=2); sp = I.solver_parameters(2, 2)
    else:
        S = I.make_solver(dim=3, resolution=int(os.environ.get("SWEEP_R", 1)), n_axons=4, degree=2 if cfg == "P2" else 1)
        sp = I.solver_parameters(3, 1)
    fields = sp._asdict(); fields.update(extra)
    if cheb >= 0:
        fields["emi_dg_chebyshev"] = bool(cheb)
    sp = namedtuple("solver_params", fields.keys())(*fields.values())
    S._unpack_solver_params(sp)
    S.verbose = False
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    return S, (E.Constant if cfg == "emix" else I.Constant)(0.0)


def run(extra):
    S, t = build(extra)
    x = S.mesh.coords[S.mesh.cells]
    d = S.mesh.gdim
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / (2.0 if d == 2 else 6.0)
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
    trace = []
    for k, ((p0, c0, m0), (p1, c1, m1)) in enumerate(zip(h, ref)):
        mem = np.nonzero(m1)[0]
        e = [np.abs(p0 - p1).max() / np.abs(p1).max(), np.abs(c0 - c1).max() / np.abs(c1).max(),
             np.abs(m0[mem] - m1[mem]).max() / np.abs(m1[mem]).max()]
        w = np.maximum(w, e)
        if (k + 1) % 5 == 0:
            trace.append("%.1e" % e[1])
    if os.environ.get("SWEEP_TRACE"):
        print("      c error every 5th step:", " ".join(trace), flush=True)
    return w


ref, its = run(dict(rtol_emi=1e-11, rtol_knp=1e-13))
print("%s cheb %d: reference run EMI %.1f / KNP %.1f its per step" % (cfg, cheb, *its), flush=True)
for th, g in pairs:
    h, its = run(dict(emi_target_safety=th or None, emi_energy_factor=g or None))          # 0: the shipped default
    print("  %s cheb %d theta %.3g g %.3g: EMI %.2f KNP %.2f its | worst phi %.2e c %.2e phi_M %.2e" % (cfg, cheb, th, g, *its, *worst(h, ref)), flush=True)
