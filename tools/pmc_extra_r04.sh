#!/bin/bash
# HBM-traffic counter passes (separate --pmc runs, no tracing next to them) of the apply kernels at the sizes the main set does not cover: r=3 and the unrefined EMIx mesh.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out
pmc() {   # name, apply_only args
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$1_$ctr -- python3 tools/apply_only.py $2 > gpurun_out/pmc_$1_$ctr.log 2>&1
    f=$(find gpurun_out/pmc_$1_$ctr -name "*counter_collection.csv" | head -1)
    python3 tools/pmc_summary.py "$f" > gpurun_out/r04_v4_pmc_$1_$ctr.txt; cat gpurun_out/r04_v4_pmc_$1_$ctr.txt
    rm -rf gpurun_out/pmc_$1_$ctr
  done
}
pmc r3 "3 3"
pmc emix0 "emix0 5"
