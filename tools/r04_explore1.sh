#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
tools/ab.sh r04_ab1.txt "KNP_NOP=1" "KNP_KNP_MIN_IT=3" "KNP_KNP_MIN_IT=1" "KNP_EXTRAPOLATE_ORDER=2" "KNP_KNP_MIN_IT=3 KNP_EXTRAPOLATE_ORDER=2" "KNP_KNP_KRYLOV=gmres" "KNP_KNP_KRYLOV=gmres KNP_KNP_MIN_IT=3" "KNP_EMI_CHEB=1"
for e in "KNP_NOP=1" "KNP_KNP_MIN_IT=3" "KNP_KNP_MIN_IT=1" "KNP_KNP_MIN_IT=3 KNP_EXTRAPOLATE_ORDER=2"; do
  echo "== $e" >> gpurun_out/r04_tol1.txt
  env $e python tools/tolerance_sweep.py 2 40 1e-5/1e-7 >> gpurun_out/r04_tol1.txt 2>&1
  tail -3 gpurun_out/r04_tol1.txt
done
