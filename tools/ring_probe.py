"""Where the ring-staged KNP apply spends its time (timing probes of csrc/apply_ring.hip, results of the probed launches are wrong by
construction): KNP_RING_DEBUG bit 0 = halo rows replaced by the block's own rows (no gather), bit 1 = consumers skip the facet terms."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd")); sys.path.insert(0, os.path.join(ROOT, "examples", "idealized_geometries"))
from idealized_common import make_solver
from knpemidg import _abi as A
r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S = make_solver(dim=3, resolution=r, degree=1)
dev = S.dev
rng = np.random.default_rng(0)
dev.upload(A.F_X, rng.uniform(-1, 1, size=dev.size(A.F_X)))
dev.upload(A.F_PHI, 0.07 * rng.uniform(-1, 1, size=dev.size(A.F_PHI)))
dev.update_kappa(); dev.update_dnphi()
nc = dev.nc_owned
meta = dev.debug_table(A.DT_META)
print("cells %d hb_stride %d" % (nc, meta[4]))
for name, env in [("ring, species split", {}), ("ring, one consumer group", {"KNP_RING_SPLIT": "0"}),
                  ("ring split, no gather", {"KNP_RING_DEBUG": "1"}), ("ring split, no facet terms", {"KNP_RING_DEBUG": "2"}),
                  ("ring split, no gather, no facet terms", {"KNP_RING_DEBUG": "3"}), ("emi staged (KNP_EMI_RING=0)", {"KNP_EMI_RING": "0"}),
                  ("ring split, consumers alone", {"KNP_RING_DEBUG": "4"}), ("ring one group, consumers alone", {"KNP_RING_SPLIT": "0", "KNP_RING_DEBUG": "4"}),
                  ("ring split, consumers alone, no facet terms", {"KNP_RING_DEBUG": "6"}),
                  ("ring one group, no gather", {"KNP_RING_SPLIT": "0", "KNP_RING_DEBUG": "1"}),
                  ("ring one group, no facet terms", {"KNP_RING_SPLIT": "0", "KNP_RING_DEBUG": "2"}),
                  ("halo-staged kernel", {"KNP_APPLY_RING": "0"})]:
    for k in ("KNP_RING_SPLIT", "KNP_RING_DEBUG", "KNP_APPLY_RING", "KNP_EMI_RING"):
        os.environ.pop(k, None)
    os.environ.update(env)
    k = dev.bench_apply(1, reps)
    e = dev.bench_apply(0, reps)
    print("%-40s knp %.2f us (%.0f GB/s alg)   emi %.2f us (variant %d)" % (name, k * 1e3, 217 * nc / k / 1e6, e * 1e3, dev.apply_variant(0)))
