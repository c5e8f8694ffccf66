"""Back-to-back time of the structured-mesh P1 applies with the thread-per-cell kernels (KNP_APPLY_RING=0) and the ring-staged ones."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver
from knpemidg import _abi as A
r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
S = make_solver(dim=3, resolution=r, degree=1)
dev = S.dev
rng = np.random.default_rng(0)
dev.upload(A.F_X, rng.uniform(-1, 1, size=dev.size(A.F_X)))
dev.upload(A.F_PHI, 0.07 * rng.uniform(-1, 1, size=dev.size(A.F_PHI)))
dev.update_kappa(); dev.update_dnphi()
nc = dev.nc_owned
for rep in range(2):
    for name, env in [("thread-per-cell", {"KNP_APPLY_RING": "0"}), ("ring", {"KNP_RING_SPLIT": "0"}), ("ring split", {})]:
        for k in ("KNP_RING_SPLIT", "KNP_APPLY_RING"):
            os.environ.pop(k, None)
        os.environ.update(env)
        k = dev.bench_apply(1, reps); e = dev.bench_apply(0, reps)
        print("%-16s knp %.2f us (%.3f of 8 TB/s)   emi %.2f us (%.3f)   variants %d %d" % (name, k * 1e3, 217 * nc / k / 1e6 / 8000, e * 1e3, 137 * nc / e / 1e6 / 8000,
                                                                                   dev.apply_variant(1), dev.apply_variant(0)))
