#!/bin/bash
# A/B of the captured V-cycle (hipGraph) against eager launches of the same kernels (KNP_NO_GRAPH=1), twice, on one box.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for env in "KNP_NO_GRAPH=0" "KNP_NO_GRAPH=1"; do
 for w in "--resolution 2" "--resolution 1" "--workload emix"; do
  env $env python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env','$w', round(d['ms_per_step'],3), d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])"
 done
done
done
