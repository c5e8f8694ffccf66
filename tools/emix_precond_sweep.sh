#!/bin/bash
# Preconditioner knobs on the refined EMIx reconstruction (973 k unstructured tets), where the solves take 4-6x the iterations of the
# idealized mesh: smoother degree of the aggregated levels / of the finest conforming level, strength threshold, prolongator smoothing.
# usage (GPU box): bash tools/emix_precond_sweep.sh [refine]
cd "$GRAFT_REPO_ROOT" || exit 1
ref=${1:-1}
run() {   # label, env assignments
  env $2 python3 bench.py --workload emix --refine $ref --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-44s %7.2f ms/step  EMI %5.1f its %6.1f ms  KNP %5.1f its %6.1f ms  setup %.2f s' % ('$1', d['ms_per_step'], c['emi_iters_per_step'], 1e3*c['emi_solve_s']/d['steps'], c['knp_iters_per_step'], 1e3*c['knp_solve_s']/d['steps'], c['setup_s_before_first_step']))"
}
run "default" "KNP_X=0"
run "AMG_DEGREE=2" "KNP_AMG_DEGREE=2"
run "AMG_DEGREE=3" "KNP_AMG_DEGREE=3"
run "DEGREE0_EMI=2 DEGREE0_KNP=2" "KNP_AMG_DEGREE0_EMI=2 KNP_AMG_DEGREE0_KNP=2"
run "DEGREE=2 DEGREE0=2" "KNP_AMG_DEGREE=2 KNP_AMG_DEGREE0_EMI=2 KNP_AMG_DEGREE0_KNP=2"
run "THETA=0.04" "KNP_AMG_THETA=0.04"
run "THETA=0.16" "KNP_AMG_THETA=0.16"
run "PSMOOTH_KNP=3" "KNP_AMG_PSMOOTH_KNP=3"
run "TRUNC=0.01" "KNP_AMG_TRUNC=0.01"
run "EMI_CHEB=1" "KNP_EMI_CHEB=1"
