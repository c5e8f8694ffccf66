#!/bin/bash
# KNP_KNP_EARLY on the other workloads: accuracy against tight solves (P2 r=1, EMIx) with the factor off / shipped, and the reference's
# 200-step run at r=1 and r=3.   usage (GPU box): bash tools/early_stop_sweep2.sh
cd "$GRAFT_REPO_ROOT" || exit 1
for e in 0 0.01; do
  echo "== KNP_KNP_EARLY=$e"
  echo -n "P2 r=1, 80 steps:  "; DEGREE=2 KNP_KNP_EARLY=$e python tools/tolerance_sweep.py 1 80 1e-5/1e-7 2>&1 | tail -1
  echo -n "EMIx, 100 steps:   "; KNP_KNP_EARLY=$e python tools/tolerance_emix.py 100 2>&1 | tail -1
  for r in 1 3; do (cd examples/idealized_geometries && KNP_KNP_EARLY=$e python run_3D.py $r 2.0e-2 nosave 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' '; echo; rm -rf results); done
done
