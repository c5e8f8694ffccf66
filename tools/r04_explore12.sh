#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
KNP_DEBUG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 >gpurun_out/r04_v4_p1_r2.json | grep -v "graph capture\|lambda_max" | head -30; tail -c 400 gpurun_out/r04_v4_p1_r2.json | head -c 200; echo
python -m pytest tests -m gpu -x -q -k "multirank or amg or solver or partitioned" > gpurun_out/gputests_r04_v6.log 2>&1; tail -4 gpurun_out/gputests_r04_v6.log
