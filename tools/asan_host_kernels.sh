#!/bin/bash
# AddressSanitizer + UBSan run of the host setup kernels (csrc/host_sparse.cpp; CPU only -- GPU sanitizers are not available on the pool):
# the file is compiled alone into a sanitized shared object, the package's host-kernel calls are pointed at it, and the mesh generators
# (3D r=1, 2D r=4), the Morton orders, the geometry classes and a whole EMI hierarchy build run through it.   usage: bash tools/asan_host_kernels.sh
set -e
cd "$(dirname "$0")/.."
g++ -fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -shared -fPIC -pthread knp-emi-dg_amd/csrc/host_sparse.cpp -o /tmp/libhs_asan.so
cat > /tmp/asan_host.py <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ["REPO"], "knp-emi-dg_amd"))
import numpy as np
from knpemidg import _abi
lib = C.CDLL("/tmp/libhs_asan.so")
for name, (res, args) in _abi.SIGNATURES.items():
    if name.startswith("knp_host_"):
        fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
_abi._lib = lib                                   # every host-kernel call of the package now goes to the sanitized build
os.environ["KNP_SETUP_THREADS"] = "4"
from knpemidg import mesh as M, amg
m, s, f = M.make_mesh_3D(1)
nc = m.num_cells()
co, cl = np.ascontiguousarray(m.coords), np.ascontiguousarray(m.cells, dtype=np.int32)
sc = np.empty(3)
assert lib.knp_host_cell_extent_median(nc, 4, 3, _abi._p(co, _abi._f64p), _abi._p(cl, _abi._i32p), _abi._p(sc, _abi._f64p)) == 0
o = _abi._morton_native(co, cl, sc); v = _abi._morton_native(co, None, sc)
g = _abi.geometry_classes(m, o)
cs = amg.ConformingSpace(m, f.array(), (1, 2))
lv = amg.build_emi_levels(cs, None, f.array(), (1, 2), np.random.default_rng(0).uniform(0.5, 1.5, (nc, 4)), 1.0)
big = M.BoxMesh((0, 0, 0), (1, 1, 1), 48, 48, 40)                # 2.2 M (cell, local facet) pairs: the parallel facet builder
print("parallel facet builder:", big.num_facets(), "facets")
print("levels", [l.A.shape[0] for l in lv], lv[-1].pinv.dtype, "classes", g[1].shape[0], "| 2D cells", M.make_mesh_2D(4)[0].num_cells())
PY
REPO=$PWD LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 python /tmp/asan_host.py
echo "no sanitizer report above = clean"
