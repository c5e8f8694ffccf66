#!/bin/bash
# Validation only (host-staged transport, says nothing about speed): bench.py with several ranks on the ONE GPU of the box over the
# shared-memory communicator -- the benchmarked r=2 mesh on 4 ranks, the EMIx workload on 3, the P2 configuration on 2 -- with the
# row-distributed finest conforming level (default) and, for the r=2 mesh, with the replicated one (KNP_AMG_DIST0=0).
# Iteration counts must equal the single-rank bench lines.  usage: shm_ranks_check.sh <tag>
tag=${1:-vX}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
o=gpurun_out/${tag}_shm_ranks.txt
: > $o
run() {   # name nranks env... -- bench args
    name=$1; n=$2; shift 2
    envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    echo "== $name: $n ranks, ${envs[*]} bench.py $*" >> $o
    env KNP_COMM_SHM=/knp_$$_$name "${envs[@]}" timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $((29700 + RANDOM % 200)) bench.py --gpus $n "$@" --no-cpu-baseline > gpurun_out/${tag}_shm_$name.json 2> gpurun_out/${tag}_shm_$name.err || { tail -20 gpurun_out/${tag}_shm_$name.err; return 1; }
    python - gpurun_out/${tag}_shm_$name.json >> $o <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("   n_gpus %d  parallelism %s  EMI %.2f / KNP %.2f iterations per step  (%.1f ms/step, host-staged: not a timing)" % (
    d["n_gpus"], c["parallelism"], c["emi_iters_per_step"], c["knp_iters_per_step"], d["ms_per_step"]))
PY
}
single() {   # name, bench args
    name=$1; shift
    echo "== $name: 1 rank bench.py $*" >> $o
    python bench.py "$@" --no-cpu-baseline > gpurun_out/${tag}_shm_$name.json 2> gpurun_out/${tag}_shm_$name.err || return 1
    python - gpurun_out/${tag}_shm_$name.json >> $o <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
c = d["config"]
print("   n_gpus 1  EMI %.2f / KNP %.2f iterations per step  (%.2f ms/step)" % (c["emi_iters_per_step"], c["knp_iters_per_step"], d["ms_per_step"]))
PY
}
single r2_single --steps 8 --warmup 2 && \
run r2_dist0 4 -- --steps 8 --warmup 2 && \
run r2_replicated 4 KNP_AMG_DIST0=0 -- --steps 8 --warmup 2 && \
single emix_single --workload emix --steps 8 --warmup 2 && \
run emix_dist0 3 -- --workload emix --steps 8 --warmup 2 && \
single p2_single --degree 2 --resolution 1 --steps 8 --warmup 2 && \
run p2_dist0 2 -- --degree 2 --resolution 1 --steps 8 --warmup 2
cat $o
