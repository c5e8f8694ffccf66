#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "emix or unstructured or partitioned" > gpurun_out/r04_t5.log 2>&1; tail -15 gpurun_out/r04_t5.log
grep -q "failed\|error" gpurun_out/r04_t5.log && exit 1
python bench.py --workload emix --refine 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_emix_refined_v1.json 2> gpurun_out/r04_emix_refined_v1.err; tail -2 gpurun_out/r04_emix_refined_v1.err; cat gpurun_out/r04_emix_refined_v1.json
python bench.py --workload emix --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_emix_v1.json 2> gpurun_out/r04_emix_v1.err; cat gpurun_out/r04_emix_v1.json
out=gpurun_out/r04_stop_sweep5.txt; : > $out
export SWEEP_TRACE=1
for cfg in P1 P2 emix 2D; do
  steps=25; [ $cfg = P1 ] && steps=40; [ $cfg = 2D ] && steps=40
  for cheb in 0 1; do
    python tools/stop_sweep_r04.py $cfg $cheb $steps 0/0 >> $out 2>gpurun_out/r04_stop_sweep_err.txt || tail -5 gpurun_out/r04_stop_sweep_err.txt
    tail -3 $out
  done
done
echo "== P2 with KNP_AMG_DEGREE=1" >> $out
for cheb in 0 1; do KNP_AMG_DEGREE=1 python tools/stop_sweep_r04.py P2 $cheb 25 0/0 >> $out 2>&1; tail -3 $out; done
echo "== P2 with KNP_KNP_CHEB=1" >> $out
for cheb in 0 1; do KNP_KNP_CHEB=1 python tools/stop_sweep_r04.py P2 $cheb 25 0/0 >> $out 2>&1; tail -3 $out; done
tools/ab.sh r04_ab5.txt "KNP_NOP=1"
WORKLOADS="--resolution 1 --degree 2 --steps 20 --warmup 5" tools/ab.sh r04_ab5_p2.txt "KNP_NOP=1" "KNP_AMG_DEGREE=1" "KNP_KNP_CHEB=1" "KNP_AMG_DEGREE=1 KNP_KNP_CHEB=1"
