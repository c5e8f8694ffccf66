for th in 10 20 30; do
  echo "== theta $th" 
  KNP_EMI_TARGET_SAFETY=$th python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print(d['ms_per_step'], c['emi_iters_per_step'], c['knp_iters_per_step'])"
done
python tools/stop_criterion_sweep.py 20 30
