#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
KNP_DEBUG=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep -v "graph capture\|lambda_max" | head -30
python -m pytest tests/test_gpu_parity.py -x -q -k "unstructured" 2>&1 | tail -3
