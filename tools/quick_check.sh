cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_solver.py tests/test_gpu_trajectory.py tests/test_gpu_multirank.py -x -q > gpurun_out/q_solver.log 2>&1 || { tail -30 gpurun_out/q_solver.log; exit 1; }
tail -2 gpurun_out/q_solver.log
for rep in 1 2; do
for env in "KNP_FUSE_CG_RESTRICT=1" "KNP_FUSE_CG_RESTRICT=0"; do
 for w in "--resolution 2 --steps 20 --warmup 5" "--resolution 3 --steps 8 --warmup 3"; do
  env $env python bench.py $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$env','$w', round(d['ms_per_step'],3), d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])"
 done
done
done
