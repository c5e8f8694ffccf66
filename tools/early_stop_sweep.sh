#!/bin/bash
# KNP_KNP_EARLY: a residual this factor under the tolerance ends a BiCGStab solve before its floor of four iterations.  Iterations and time of
# the reference's 200-step run (quiet after step ~45) and the accuracy of the stimulated steps against tight solves.   usage (GPU box): bash tools/early_stop_sweep.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for e in 0 0.1 0.01 0.001; do
  echo "== KNP_KNP_EARLY=$e"
  (cd examples/idealized_geometries && KNP_KNP_EARLY=$e python run_3D.py 2 2.0e-2 nosave 2>&1 | grep -v amdgpu.ids | tail -2; rm -rf results)
  KNP_KNP_EARLY=$e python tools/tolerance_sweep.py 1 100 1e-5/1e-7 2>&1 | tail -1
  KNP_KNP_EARLY=$e python tools/tolerance_sweep.py 2 100 1e-5/1e-7 2>&1 | tail -1
done
