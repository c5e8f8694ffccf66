cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_solver.py -x -q -k "amg or vcycle or precond or hierarchy or tolerances" > gpurun_out/q_solver.log 2>&1 || { tail -30 gpurun_out/q_solver.log; exit 1; }
tail -2 gpurun_out/q_solver.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/q_prof.json 2> gpurun_out/q_prof.err
find gpurun_out/prof_q -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/q_kernel_stats.csv
rm -rf gpurun_out/prof_q
grep -h "k_dense_mv" gpurun_out/q_kernel_stats.csv | cut -c1-60,100-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])"
python bench.py --workload emix --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emix', d['ms_per_step'], d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'])"
