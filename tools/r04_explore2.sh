#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/r04_stop_sweep.txt; : > $out
for cfg in P1 P2 emix 2D; do
  steps=25; [ $cfg = P1 ] && steps=40; [ $cfg = 2D ] && steps=40
  for cheb in 0 1; do
    python tools/stop_sweep_r04.py $cfg $cheb $steps 20/1 20/10 20/100 10/10 >> $out 2>gpurun_out/r04_stop_sweep_err.txt || tail -5 gpurun_out/r04_stop_sweep_err.txt
    tail -5 $out
  done
done
tools/ab.sh r04_ab2.txt "KNP_NOP=1" "KNP_EXTRAPOLATE_ORDER_KNP=2" "KNP_EMI_ENERGY_FACTOR=10" "KNP_EMI_ENERGY_FACTOR=100"
