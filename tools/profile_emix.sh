#!/bin/bash
# rocprofv3 kernel statistics of BASELINE configs[4] (EMIx tissue reconstruction) -> gpurun_out/<tag>_emix_kernel_stats.csv, <tag>_emix_profiled.json
tag=${1:-vX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_emix_$tag -- python3 bench.py --workload emix --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_emix_profiled.json 2> gpurun_out/prof_emix_$tag.err
find gpurun_out/prof_emix_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_emix_kernel_stats.csv
rm -rf gpurun_out/prof_emix_$tag
