#!/bin/bash
# usage: tools/tune_p2.sh "VAR=val VAR2=val2" ...   (each argument = one P2 bench run on the r=1 mesh, BASELINE configs[2])
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --degree 2 --resolution ${TUNE_R:-1} --no-cpu-baseline --steps ${TUNE_STEPS:-15} --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('%.2f ms/step  emi its %.1f (%.3f s)  knp its %.1f (%.3f s)' % (d['ms_per_step'], c['emi_iters_per_step'], c['emi_solve_s'], c['knp_iters_per_step'], c['knp_solve_s']))"
done
