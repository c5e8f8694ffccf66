#!/bin/bash
# rocprofv3 kernel statistics of the r=3 mesh (7.96 M tets, 95.6 M DoFs) -> gpurun_out/<tag>_bench_r3_kernel_stats.csv
tag=${1:-vX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_$tag -- python3 bench.py --resolution 3 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_r3_profiled.json 2> gpurun_out/prof_r3_$tag.err
find gpurun_out/prof_r3_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_bench_r3_kernel_stats.csv
rm -rf gpurun_out/prof_r3_$tag
