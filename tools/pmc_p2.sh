#!/bin/bash
# HBM-side traffic of the P2 apply kernels: separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/apply_only.py <r> 5 2
r=${1:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_p2_$ctr -- python3 tools/apply_only.py $r 5 2 > gpurun_out/pmc_p2_r${r}_$ctr.log 2>&1
  f=$(find gpurun_out/pmc_p2_$ctr -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" > gpurun_out/pmc_p2_r${r}_$ctr.txt; cat gpurun_out/pmc_p2_r${r}_$ctr.txt
  rm -rf gpurun_out/pmc_p2_$ctr
done
