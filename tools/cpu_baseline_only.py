"""bench.py's CPU baseline alone (no GPU work): python tools/cpu_baseline_only.py [resolution]   (KNP_CPU_BASELINE_CORES / _BLAS as in bench.py)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

if __name__ == "__main__":
    out = bench.cpu_baseline(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    print(json.dumps({k: v for k, v in out.items() if k != "sample"}))
    print(out["sample"])
