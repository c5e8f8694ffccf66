#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python tools/profile_setup.py 2 > gpurun_out/r04_setup_profile_r2_v2.txt 2>&1; grep -v "^$" gpurun_out/r04_setup_profile_r2_v2.txt | head -64
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_v2_p1_r2.json 2> gpurun_out/r04_v2_p1_r2.err; cat gpurun_out/r04_v2_p1_r2.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_v2_p1_r2.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'], d['config']['emi_dg_smoother'], d['config']['setup_s_before_first_step'])"
python bench.py > gpurun_out/r04_v2_bench_default.json 2> gpurun_out/r04_v2_bench_default.err; cat gpurun_out/r04_v2_bench_default.err; tail -c 1500 gpurun_out/r04_v2_bench_default.json
python bench.py --workload emix --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -c 600
python -m pytest tests -m gpu -x -q -k "tables or multirank or golden or amg" > gpurun_out/gputests_r04_v3.log 2>&1; tail -4 gpurun_out/gputests_r04_v3.log
