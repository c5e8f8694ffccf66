#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
out=gpurun_out/r04_stop_sweep4.txt; : > $out
export SWEEP_TRACE=1 KNP_KNP_MIN_IT=4
for cfg in P2 P1 emix; do
  steps=25; [ $cfg = P1 ] && steps=40
  for cheb in 0 1; do
    python tools/stop_sweep_r04.py $cfg $cheb $steps 1e9/0.3 1e9/0.1 1e9/0.03 1e9/0.01 >> $out 2>gpurun_out/r04_stop_sweep_err.txt || tail -5 gpurun_out/r04_stop_sweep_err.txt
    tail -9 $out
  done
done
python tools/stop_sweep_r04.py 2D 0 40 1e9/1 1e9/0.3 1e9/0.1 >> $out 2>&1; tail -7 $out
python tools/stop_sweep_r04.py 2D 1 40 1e9/1 1e9/0.3 1e9/0.1 >> $out 2>&1; tail -7 $out
# the refined EMIx workload on the existing coordinate-path kernels, and the host-setup profile
python bench.py --workload emix --refine 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_emix_refined_v0.json 2> gpurun_out/r04_emix_refined_v0.err; tail -3 gpurun_out/r04_emix_refined_v0.err; cat gpurun_out/r04_emix_refined_v0.json
python tools/profile_setup.py 2 > gpurun_out/r04_setup_profile_r2.txt 2>&1; grep -v "^$" gpurun_out/r04_setup_profile_r2.txt | head -70
