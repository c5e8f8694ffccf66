#!/bin/bash
# rocprofv3 kernel statistics of the default workload (r=2, 20 steps) + the HBM-traffic counter passes of the apply kernels
# usage (on the GPU box): bash tools/profile_r2.sh [tag]   -> gpurun_out/<tag>_bench_r2_kernel_stats.csv, <tag>_bench_r2_profiled.json, pmc_t_*.txt
tag=${1:-r02_v3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v3 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_r2_profiled.json 2> gpurun_out/prof_v3.err
find gpurun_out/prof_v3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_bench_r2_kernel_stats.csv
rm -rf gpurun_out/prof_v3
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_t_$ctr -- python3 tools/apply_only.py 2 5 > gpurun_out/pmc_t_$ctr.log 2>&1
  f=$(find gpurun_out/pmc_t_$ctr -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" > gpurun_out/pmc_t_$ctr.txt; cat gpurun_out/pmc_t_$ctr.txt
  rm -rf gpurun_out/pmc_t_$ctr
done
