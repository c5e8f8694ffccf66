#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputests_r04_v2.log 2>&1; tail -8 gpurun_out/gputests_r04_v2.log
tools/ab.sh r04_ab6.txt "KNP_NOP=1" "KNP_EMI_CHEB=0" "KNP_EMI_CHEB=1" "KNP_KNP_MIN_IT=3"
WORKLOADS="--resolution 1 --degree 2 --steps 20 --warmup 5;--resolution 2 --degree 2 --steps 10 --warmup 5" tools/ab.sh r04_ab6_p2.txt "KNP_NOP=1"
