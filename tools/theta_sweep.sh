#!/bin/bash
# Strength-of-connection threshold of the smoothed aggregation (KNP_AMG_THETA) on the bench workloads.   usage (GPU box): bash tools/theta_sweep.sh
cd "$GRAFT_REPO_ROOT" || exit 1
run() {   # label, env, bench args
  env $2 python3 bench.py $3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-34s %7.3f ms/step  EMI %5.2f its %6.2f ms  KNP %5.2f its %6.2f ms  setup %.2f s' % ('$1', d['ms_per_step'], c['emi_iters_per_step'], 1e3*c['emi_solve_s']/d['steps'], c['knp_iters_per_step'], 1e3*c['knp_solve_s']/d['steps'], c['setup_s_before_first_step']))"
}
for th in 0.08 0.04 0.02; do
  run "r2 theta=$th" "KNP_AMG_THETA=$th" "--steps 20 --warmup 5"
  run "emix theta=$th" "KNP_AMG_THETA=$th" "--workload emix --steps 20 --warmup 5"
  run "emix refined theta=$th" "KNP_AMG_THETA=$th" "--workload emix --refine 1 --steps 10 --warmup 3"
  run "P2 r1 theta=$th" "KNP_AMG_THETA=$th" "--degree 2 --resolution 1 --steps 20 --warmup 5"
done
run "r3 theta=0.08" "KNP_AMG_THETA=0.08" "--resolution 3 --steps 8 --warmup 4"
run "r3 theta=0.04" "KNP_AMG_THETA=0.04" "--resolution 3 --steps 8 --warmup 4"
