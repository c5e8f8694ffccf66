#!/bin/bash
# The bench lines and the rocprofv3 summary committed under profiles/ for a version tag ($1), on the GPU box.
tag=${1:-vX}
cd "$GRAFT_REPO_ROOT" || exit 1
o=gpurun_out
python bench.py > $o/${tag}_bench_default.json 2> $o/${tag}_bench_default.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_p1_r2.json 2> $o/${tag}_p1_r2.err || exit 1
python bench.py --resolution 3 --steps 10 --warmup 3 --no-cpu-baseline > $o/${tag}_p1_r3.json 2> $o/${tag}_p1_r3.err || exit 1
python bench.py --degree 2 --resolution 1 --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_p2_r1.json 2> $o/${tag}_p2_r1.err || exit 1
python bench.py --degree 2 --resolution 2 --steps 10 --warmup 3 --no-cpu-baseline > $o/${tag}_p2_r2.json 2> $o/${tag}_p2_r2.err || exit 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_bench_r2_profiled.json 2> $o/prof_$tag.err
find $o/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $o/${tag}_bench_r2_kernel_stats.csv
rm -rf $o/prof_$tag
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("bench_default", "p1_r2", "p1_r3", "p2_r1", "p2_r2", "bench_r2_profiled"):
    try:
        d = json.loads(open("gpurun_out/%s_%s.json" % (tag, f)).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "failed", e); continue
    r = d["roofline"]
    print(f, "%.2f ms/step %.3f GDoF/s its %.2f/%.2f | %s frac %.3f in-solver %.1f b2b %.1f | emi frac %.3f in %.1f b2b %.1f" % (
        d["ms_per_step"], d["value"] / 1e9, d["config"]["emi_iters_per_step"], d["config"]["knp_iters_per_step"], r["kernel"], r["frac"],
        r["in_solver_us"], r["back_to_back_us"], r["emi_apply"]["frac"], r["emi_apply"]["in_solver_us"], r["emi_apply"]["back_to_back_us"]))
PY
grep -h "bench" $o/${tag}_p1_r3.err $o/${tag}_bench_default.err $o/${tag}_p2_r2.err | grep -v step
