#!/bin/bash
# A/B runs of bench.py on the GPU box: one line per (environment, workload) with ms/step, iterations per step and the in-solver apply
# times.   usage: tools/ab.sh OUT "ENV1=.. ENV2=.." "ENVA=.." ...      (WORKLOADS overrides the workload list, ';'-separated)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out="gpurun_out/$1"; shift
: > "$out"
IFS=';' read -ra wl <<< "${WORKLOADS:---resolution 2 --steps 20 --warmup 5}"
for envs in "$@"; do
  for w in "${wl[@]}"; do
    env $envs python bench.py $w --no-cpu-baseline 2>gpurun_out/ab_last_stderr.log | ENVS="$envs" W="$w" python -c "
import sys, json, os
lines = [l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')]
if not lines:
    print(os.environ['ENVS'], '|', os.environ['W'], '| FAILED'); sys.exit(0)
d = json.loads(lines[-1]); r = d['roofline']
print('%-60s | %-40s | %7.3f ms/step  its %.2f / %.2f  knp apply %.1f us (%.3f)  emi apply %.1f us (%.3f)' % (os.environ['ENVS'], os.environ['W'],
      d['ms_per_step'], d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'], r['in_solver_us'], r['frac'],
      r['emi_apply']['in_solver_us'], r['emi_apply']['frac']))" >> "$out"
    tail -1 "$out"
  done
done
