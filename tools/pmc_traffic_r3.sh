cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_r3_$ctr -- python3 tools/apply_only.py 3 3 > gpurun_out/pmc_r3_$ctr.log 2>&1
  f=$(find gpurun_out/pmc_r3_$ctr -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py "$f" > gpurun_out/pmc_r3_$ctr.txt; cat gpurun_out/pmc_r3_$ctr.txt
  rm -rf gpurun_out/pmc_r3_$ctr
done
