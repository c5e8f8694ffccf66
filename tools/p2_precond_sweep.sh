#!/bin/bash
# Preconditioner knobs of the DG-P2 configuration (configs[2]: r=1, 3.73 M DoFs).   usage (GPU box): bash tools/p2_precond_sweep.sh [resolution]
cd "$GRAFT_REPO_ROOT" || exit 1
r=${1:-1}
run() {
  env $2 python3 bench.py --degree 2 --resolution $r --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-40s %7.3f ms/step  EMI %5.2f its %6.2f ms  KNP %5.2f its %6.2f ms' % ('$1', d['ms_per_step'], c['emi_iters_per_step'], 1e3*c['emi_solve_s']/d['steps'], c['knp_iters_per_step'], 1e3*c['knp_solve_s']/d['steps']))"
}
run "default" "KNP_X=0"
run "TOPDEGREE=1" "KNP_AMG_TOPDEGREE=1"
run "TOPDEGREE=3" "KNP_AMG_TOPDEGREE=3"
run "TOPLOWER=0.2" "KNP_AMG_TOPLOWER=0.2"
run "TOPLOWER=0.05" "KNP_AMG_TOPLOWER=0.05"
run "AMG_DEGREE=2" "KNP_AMG_DEGREE=2"
run "PSMOOTH_EMI=2" "KNP_AMG_PSMOOTH_EMI=2"
run "EMI_CHEB=1" "KNP_EMI_CHEB=1"
run "EMI_CHEB=1 TOPDEGREE=1" "KNP_EMI_CHEB=1 KNP_AMG_TOPDEGREE=1"
run "ENERGY_FACTOR=1.0" "KNP_EMI_ENERGY_FACTOR=1.0"
