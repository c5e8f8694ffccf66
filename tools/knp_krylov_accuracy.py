"""Worst errors of the production run against the oracle's trajectories (tests/golden/traj_*_P1.npz: direct solves + LSODA) for a KNP
Krylov configuration given by the environment (KNP_KNP_KRYLOV, KNP_GMRES_TRUNC, KNP_D8_FACTOR).  Prints per trajectory the worst
relative errors over all steps and the KNP iteration counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("knp-emi-dg_amd", "tests", "oracle", os.path.join("examples", "idealized_geometries")):
    sys.path.insert(0, os.path.join(ROOT, p))
import test_gpu_trajectory as T
from idealized_common import Constant
for name, dim in (("traj_2D_r2_P1", 2), ("traj_3D_r0_4axon_P1", 3)):
    g = np.load(os.path.join(T.GOLD, name + ".npz"))
    S = T._solver(dim, 1, tight=False)
    vol = T._cell_volumes(S.mesh)
    t = Constant(0.0)
    worst = {}
    for k in range(int(g["n_steps"])):
        S.step_membrane_models(k)
        S.solve_for_time_step(k, t)
        e = T._errors(S, g, k, vol)
        for key, v in e.items():
            worst[key] = max(worst.get(key, 0.0), float(v))
    print(name, {k: "%.2e" % v for k, v in worst.items()}, "KNP its/step %.2f" % np.mean([max(n) for n in S.knp_niter]), "EMI %.2f" % np.mean(S.emi_niter))
    S.dev.close()
