#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
WORKLOADS="--resolution 2 --steps 20 --warmup 5;--workload emix --steps 20 --warmup 5;--workload emix --refine 1 --steps 10 --warmup 3;--resolution 1 --degree 2 --steps 20 --warmup 5;--resolution 1 --steps 20 --warmup 5" tools/ab.sh r04_ab9.txt "KNP_NOP=1"
for w in "--resolution 2 --steps 20 --warmup 5" "--workload emix --steps 20 --warmup 5"; do python bench.py $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['emi_dg_smoother'], d['config']['setup_s_before_first_step'])"; done
python -m pytest tests -m gpu -x -q -k "multirank or production or config" > gpurun_out/gputests_r04_v4.log 2>&1; tail -4 gpurun_out/gputests_r04_v4.log
