"""Runs only the operator-apply kernels on the r=R idealized 3D mesh (profiling helper)."""
import os, sys, time
import numpy as np
os.environ.setdefault("KNP_AMG_SERIAL_SETUP", "1")     # no preconditioner is built here: do not start the setup helper processes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver
from knpemidg import _abi as A
# usage: apply_only.py R REPS [DEGREE]        idealized 4-axon mesh at refinement R
#        apply_only.py emixN REPS [DEGREE]    the EMIx tissue reconstruction after N regular refinements (emix0: 121 617 tets, emix1: 972 936)
r = sys.argv[1] if len(sys.argv) > 1 else "2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if r.startswith("emix"):
    sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
    import emix_common
    S = emix_common.make_solver(degree=degree, refine=int(r[4:] or 0))
else:
    S = make_solver(dim=3, resolution=int(r), degree=degree)
dev = S.dev
rng = np.random.default_rng(0)
dev.upload(A.F_X, rng.uniform(-1, 1, size=dev.size(A.F_X)))
dev.upload(A.F_PHI, (70.0 if r.startswith("emix") else 0.07) * rng.uniform(-1, 1, size=dev.size(A.F_PHI)))
dev.update_kappa(); dev.update_dnphi()
e = dev.bench_apply(0, reps); k = dev.bench_apply(1, reps)
nc = dev.nc_owned
be, bk = (137, 217) if degree == 1 else (281, 457)
print("cells %d  emi %.2f us (%.0f GB/s alg)  knp %.2f us (%.0f GB/s alg)" % (nc, e * 1e3, be * nc / e / 1e6, k * 1e3, bk * nc / k / 1e6))
