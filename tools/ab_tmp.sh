set -e
python -m pytest tests/test_gpu_solver.py tests/test_gpu_trajectory.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1 || (tail -40 gpurun_out/r3_t2.log; exit 1)
tail -3 gpurun_out/r3_t2.log
rm -f gpurun_out/r3_ab2.log
for cfg in "KNP_BJ_TABLE=0" "KNP_BJ_TABLE=1"; do
  for args in "--resolution 2 --degree 1" "--resolution 1 --degree 2"; do
  echo "== $cfg $args" >> gpurun_out/r3_ab2.log
  env $cfg KNP_DEBUG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $args 2> gpurun_out/r3_ab2.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print(d['ms_per_step'], c['emi_iters_per_step'], c['knp_iters_per_step'], d['roofline']['in_solver_us'], d['roofline']['emi_apply']['in_solver_us'])" >> gpurun_out/r3_ab2.log
  grep "table" gpurun_out/r3_ab2.err >> gpurun_out/r3_ab2.log || true
  done
done
cat gpurun_out/r3_ab2.log
