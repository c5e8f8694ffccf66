#!/bin/bash
# usage: tools/tune_p1.sh "VAR=val VAR2=val2" ...   (each argument = one P1 bench run on the r=2 mesh, the headline workload)
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --resolution ${TUNE_R:-2} --no-cpu-baseline --steps ${TUNE_STEPS:-20} --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('%.2f ms/step  emi its %.2f (%.3f s)  knp its %.2f (%.3f s)' % (d['ms_per_step'], c['emi_iters_per_step'], c['emi_solve_s'], c['knp_iters_per_step'], c['knp_solve_s']))"
done
