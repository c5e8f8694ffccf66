#!/bin/bash
# rocprofv3 kernel statistics of the P2 configuration (configs[2]: --degree 2 --resolution 1) -> gpurun_out/<tag>_bench_p2r1_kernel_stats.csv
tag=${1:-vX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p2_$tag -- python3 bench.py --degree 2 --resolution 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_p2r1_profiled.json 2> gpurun_out/prof_p2_$tag.err
find gpurun_out/prof_p2_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_bench_p2r1_kernel_stats.csv
rm -rf gpurun_out/prof_p2_$tag
