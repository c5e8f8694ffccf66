#!/bin/bash
# The per-rank problem size of the 8-GPU strong-scaling run (r=2 mesh / 8 = the r=1 mesh, 124 416 tets) on one GPU: bench line and
# rocprofv3 kernel statistics -- what the DG-level and V-cycle kernels cost in the launch-latency regime (DESIGN.md §6 model).
tag=${1:-vX}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python bench.py --resolution 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_p1_r1.json 2> gpurun_out/${tag}_p1_r1.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1_$tag -- python3 bench.py --resolution 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_r1_profiled.json 2> gpurun_out/prof_r1_$tag.err
find gpurun_out/prof_r1_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${tag}_bench_r1_kernel_stats.csv
rm -rf gpurun_out/prof_r1_$tag
python - "$tag" <<'PY'
import json, sys, csv
tag = sys.argv[1]
d = json.loads(open("gpurun_out/%s_p1_r1.json" % tag).read().strip().splitlines()[-1])
r = d["roofline"]
print("r=1: %.3f ms/step, its %.2f / %.2f, %s in-solver %.1f us b2b %.1f, emi %.1f / %.1f" % (d["ms_per_step"], d["config"]["emi_iters_per_step"],
      d["config"]["knp_iters_per_step"], r["kernel"], r["in_solver_us"], r["back_to_back_us"], r["emi_apply"]["in_solver_us"], r["emi_apply"]["back_to_back_us"]))
rows = list(csv.DictReader(open("gpurun_out/%s_bench_r1_kernel_stats.csv" % tag)))
tot = sum(int(x["TotalDurationNs"]) for x in rows)
print("kernel time per step %.2f ms, launches per step %.0f" % (tot / 25e6, sum(int(x["Calls"]) for x in rows) / 25))
for x in rows[:14]:
    print("%6.2f%% %5d %7.1f us %s" % (float(x["Percentage"]), int(x["Calls"]), float(x["AverageNs"]) / 1e3, x["Name"][:90]))
PY
