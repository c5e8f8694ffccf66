#!/bin/bash
# The reference's own 3D run (examples/idealized-geometries/run_3D.py: 200 steps of 0.1 ms, fields and solver statistics saved) on this
# build, end to end, at resolution $1 (default 2); wall time and the result file's datasets go to gpurun_out/run3d_r$1.log
r=${1:-2}
cd "$GRAFT_REPO_ROOT/examples/idealized_geometries" || exit 1
rm -rf results
start=$(date +%s.%N)
python run_3D.py $r 2.0e-2 > "$GRAFT_REPO_ROOT/gpurun_out/run3d_r$r.out" 2>&1 || { tail -5 "$GRAFT_REPO_ROOT/gpurun_out/run3d_r$r.out"; exit 1; }
end=$(date +%s.%N)
python - "$r" "$start" "$end" <<'PY' > "$GRAFT_REPO_ROOT/gpurun_out/run3d_r$r.log"
import sys, os, glob
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "knp-emi-dg_amd"))
import numpy as np
from knpemidg.h5lite import H5File
r, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
print("run_3D.py %s 2.0e-2: %.1f s wall (process start to exit)" % (r, t1 - t0))
for f in sorted(glob.glob("results/data/3D/*")):
    print(" ", f, os.path.getsize(f), "bytes")
h5 = [f for f in glob.glob("results/data/3D/*.h5")]
if h5:
    H = H5File(h5[0])
    names = sorted(H.datasets)
    print("datasets:", len(names), "first:", names[:6], "last:", names[-3:])
    pots = [n for n in names if n.startswith("potential/vector_")]
    last = max(pots, key=lambda n: int(n.rsplit("_", 1)[1]))
    phi = np.asarray(H.read(last)).ravel()
    print(last, "min %.6e max %.6e finite %s" % (phi.min(), phi.max(), np.isfinite(phi).all()))
for f in glob.glob("results/data/3D/solver/*niter*") + glob.glob("results/data/3D/*niter*"):
    vals = []
    for line in open(f):
        try:
            vals.append(float(line.split()[-1]))
        except (ValueError, IndexError):
            pass
    if vals:
        print(" ", os.path.basename(f), "n=%d mean %.2f max %d" % (len(vals), np.mean(vals), max(vals)))
PY
cat "$GRAFT_REPO_ROOT/gpurun_out/run3d_r$r.log"
