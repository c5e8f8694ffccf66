#!/bin/bash
# DG-P2 with the DG-level Chebyshev steps by the new default rule: accuracy (P2 r=1, 40 steps) + tests + bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out

timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_trajectory.py -x -q > gpurun_out/q_p2.log 2>&1; tail -3 gpurun_out/q_p2.log
for w in "--resolution 1 --steps 20 --warmup 5" "--resolution 2 --steps 10 --warmup 3"; do
  python bench.py --degree 2 $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default','$w', round(d['ms_per_step'],3), d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])"
done
