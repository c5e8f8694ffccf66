#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputests_r04_v5.log 2>&1; tail -5 gpurun_out/gputests_r04_v5.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_v3_p1_r2.json 2> gpurun_out/r04_v3_p1_r2.err; cat gpurun_out/r04_v3_p1_r2.err
python bench.py --resolution 3 --steps 8 --warmup 5 --no-cpu-baseline 2>&1 >/dev/null | tail -5
python tools/profile_setup.py 2 2>&1 | grep -v "^$" | head -24
