#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/r04_stop_sweep3.txt; : > $out
export SWEEP_TRACE=1
for d8 in 20 10 5; do
  for cfg in P1 P2; do
    steps=25; [ $cfg = P1 ] && steps=40
    for cheb in 0 1; do
      echo "== KNP_D8_FACTOR=$d8" >> $out
      KNP_D8_FACTOR=$d8 python tools/stop_sweep_r04.py $cfg $cheb $steps 20/1 10/1 >> $out 2>gpurun_out/r04_stop_sweep_err.txt || tail -5 gpurun_out/r04_stop_sweep_err.txt
      tail -6 $out
    done
  done
done
echo "== min_it 4 / 5, d8 20" >> $out
for mi in 4 5; do
  KNP_KNP_MIN_IT=$mi python tools/stop_sweep_r04.py P1 0 40 20/1 >> $out 2>&1; tail -3 $out
  KNP_KNP_MIN_IT=$mi python tools/stop_sweep_r04.py P2 1 25 20/1 >> $out 2>&1; tail -3 $out
done
