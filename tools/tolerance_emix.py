"""Per-step accuracy of the shipped tolerances on the EMIx configuration (BASELINE configs[4], one GPU) against the same run converged
to 1e-11 / 1e-13.  usage: tolerance_emix.py [steps] [emi_rtol_scale,rtol_knp ...]   (default: the example's shipped parameters)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "emix_simulations")]
import emix_common as E
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def run(tight, override=None):
    S = E.make_solver()
    sp = E.solver_parameters()
    if tight:
        sp = sp._replace(rtol_emi=1e-11, rtol_knp=1e-13)
    elif override is not None:
        from collections import namedtuple
        fields = dict(sp._asdict(), emi_rtol_scale=override[0], rtol_knp=override[1], knp_rtol_scale=1.0)   # the round-2 test with explicit factors
        sp = namedtuple("solver_params", fields.keys())(*fields.values())
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = E.Constant(0.0)
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


ref, its_ref = run(True)
cases = [tuple(float(x) for x in a.split(",")) for a in sys.argv[2:]] or [None]
for case in cases:
    h, its = run(False, case)
    print("case", case, "EMI its mean %.1f KNP its mean %.1f" % (np.mean(its[0]), np.mean(its[1])))
    worst = np.zeros(3)
    for k in range(steps):
        (p0, c0, m0), (p1, c1, m1) = h[k], ref[k]
        mem = np.nonzero(m1)[0]
        e = np.array([np.abs(p0 - p1).max() / np.abs(p1).max(), np.abs(c0 - c1).max() / np.abs(c1).max(),
                      np.abs(m0[mem] - m1[mem]).max() / np.abs(m1[mem]).max()])
        worst = np.maximum(worst, e)
    print("   worst over %d steps: phi %.2e   c %.2e   phi_M %.2e" % (steps, *worst))
