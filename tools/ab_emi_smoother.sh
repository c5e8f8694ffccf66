#!/bin/bash
# DG-level Chebyshev step of the EMI preconditioner: default (decided per mesh by Solver._emi_dg_chebyshev) against forced on / off
# -> gpurun_out/r03_emi_dg_smoother.txt (committed as profiles/r03_emi_dg_smoother.txt together with the accuracy runs named there)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
o=gpurun_out/r03_emi_dg_smoother.txt
echo "# tools/ab_emi_smoother.sh: bench.py --steps 20 --warmup 5 (r=3: 8 + 3) with the DG-level Chebyshev step of the EMI preconditioner by default rule, forced on (KNP_EMI_CHEB=1) and forced off (KNP_EMI_CHEB=0); ms/step, EMI / KNP iterations per step" > $o
for env in "KNP_DEBUG=0" "KNP_EMI_CHEB=1" "KNP_EMI_CHEB=0"; do
 for w in "--resolution 2" "--resolution 1" "--workload emix" "--resolution 3 --steps 8 --warmup 3"; do
  env $env python bench.py --steps 20 --warmup 5 $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env'.replace('KNP_DEBUG=0','default      '),'$w', round(d['ms_per_step'],3), d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])" >> $o
 done
done
cat $o
