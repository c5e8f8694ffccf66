#!/bin/bash
# Round-4 evidence set (on the GPU box): rocprofv3 kernel statistics of the bench workloads + the HBM-traffic counter passes of the
# apply kernels (separate --pmc runs, no tracing next to them).   usage: bash tools/final_profiles_r04.sh TAG
tag=${1:-r04_v1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
prof() {   # name, bench args
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py $2 --no-cpu-baseline > gpurun_out/${tag}_$1_profiled.json 2> gpurun_out/prof_$1.err
  find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_$1_kernel_stats.csv
  rm -rf gpurun_out/prof_$1
  tail -c 600 gpurun_out/${tag}_$1_profiled.json | head -c 300; echo
}
prof bench_r2 "--steps 20 --warmup 5"
prof p2_r1 "--degree 2 --resolution 1 --steps 20 --warmup 5"
prof emix_refined "--workload emix --refine 1 --steps 10 --warmup 3"
pmc() {   # name, apply_only args
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$1_$ctr -- python3 tools/apply_only.py $2 > gpurun_out/pmc_$1_$ctr.log 2>&1
    f=$(find gpurun_out/pmc_$1_$ctr -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summary.py "$f" > gpurun_out/${tag}_pmc_$1_$ctr.txt; cat gpurun_out/${tag}_pmc_$1_$ctr.txt
    rm -rf gpurun_out/pmc_$1_$ctr
  done
}
pmc r2 "2 5"
pmc p2r1 "1 5 2"
pmc p2r2 "2 5 2"
pmc emix1 "emix1 5"
# un-profiled bench lines of the same commit
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_p1_r2.json 2>gpurun_out/${tag}_p1_r2.err
python3 bench.py > gpurun_out/${tag}_bench_default.json 2>gpurun_out/${tag}_bench_default.err
python3 bench.py --resolution 3 --steps 8 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_p1_r3.json 2>gpurun_out/${tag}_p1_r3.err
python3 bench.py --degree 2 --resolution 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_p2_r1.json 2>/dev/null
python3 bench.py --degree 2 --resolution 2 --steps 10 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_p2_r2.json 2>/dev/null
python3 bench.py --workload emix --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_emix.json 2>/dev/null
python3 bench.py --workload emix --refine 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_emix_refined.json 2>/dev/null
for f in p1_r2 bench_default p1_r3 p2_r1 p2_r2 emix emix_refined; do python3 -c "
import json,sys
d=json.loads(open('gpurun_out/${tag}_$f.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$f', round(d['ms_per_step'],3), 'ms/step', d['config']['emi_iters_per_step'], d['config']['knp_iters_per_step'], r['kernel'], round(r['frac'],3), r['emi_apply']['kernel'], round(r['emi_apply']['frac'],3), d['config'].get('emi_dg_smoother'))"; done
