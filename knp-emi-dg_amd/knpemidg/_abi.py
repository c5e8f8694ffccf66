"""ctypes binding of libknpemi_hip.so (the C ABI declared in include/knpemi_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no
MI355X is visible when a context is created, this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KNP_LIB_PATH: another build of the same library (A/B runs of a kernel variant, tools/build_variant.sh); the product is the in-tree one
LIB_PATH = os.environ.get("KNP_LIB_PATH") or os.path.join(_HERE, "libknpemi_hip.so")

# enum knp_field (include/knpemi_hip.h)
F_PHI, F_C, F_C_PREV, F_C_ELIM, F_PHI_M, F_I_CH, F_E, F_KAPPA, F_DNPHI, F_B_EMI, F_B_KNP, F_X, F_Y, \
    F_FACET_TMP = range(14)

FACET_TMP_SLOTS = 4        # KNP_FACET_TMP_SLOTS (include/knpemi_hip.h)
# knp_debug_table_id (include/knpemi_hip.h)
DT_CELLS, DT_NBR, DT_FLAG, DT_CFACET, DT_MF, DT_HB_SRC, DT_HB_LOC, DT_META = range(8)

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_u32p = C.POINTER(C.c_uint32)
_i8p = C.POINTER(C.c_int8)
_f32p = C.POINTER(C.c_float)
_ctxp = C.c_void_p

# every exported symbol of include/knpemi_hip.h with its signature
SIGNATURES = {
    "knp_ctx_create": (C.c_int, [C.POINTER(_ctxp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                 C.c_int64, C.c_int64, _f64p, _i32p, _u32p, _i32p, _i8p, _u32p, C.c_int, _u32p]),
    "knp_ctx_destroy": (None, [_ctxp]),
    "knp_last_error": (C.c_char_p, [_ctxp]),
    "knp_set_params": (C.c_int, [_ctxp] + [C.c_double] * 8 + [_f64p, _f64p, _f64p, _f64p, C.c_int]),
    "knp_set_geometry_classes": (C.c_int, [_ctxp, C.c_int, C.POINTER(C.c_uint16), _f64p]),
    "knp_set_tabulation": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_int, _f64p, _f64p, _f64p]),
    "knp_set_mms": (C.c_int, [_ctxp, _f64p, _f64p, _f64p]),
    "knp_set_source": (C.c_int, [_ctxp, _f64p]),
    "knp_field_size": (C.c_int64, [_ctxp, C.c_int]),
    "knp_debug_table_size": (C.c_int64, [_ctxp, C.c_int]),
    "knp_debug_table": (C.c_int, [_ctxp, C.c_int, C.c_void_p, C.c_int64]),
    "knp_upload": (C.c_int, [_ctxp, C.c_int, _f64p, C.c_int64, C.c_int64]),
    "knp_download": (C.c_int, [_ctxp, C.c_int, _f64p, C.c_int64, C.c_int64]),
    "knp_copy_field": (C.c_int, [_ctxp, C.c_int, C.c_int]),
    "knp_update_kappa": (C.c_int, [_ctxp]),
    "knp_update_dnphi": (C.c_int, [_ctxp]),
    "knp_emi_apply": (C.c_int, [_ctxp, C.c_int, C.c_int]),
    "knp_knp_apply": (C.c_int, [_ctxp, C.c_int, C.c_int]),
    "knp_emi_rhs": (C.c_int, [_ctxp]),
    "knp_knp_rhs": (C.c_int, [_ctxp]),
    "knp_emi_residual_target": (C.c_int, [_ctxp, C.c_double]),
    "knp_knp_load_measure": (C.c_int, [_ctxp, _f64p]),
    "knp_knp_early_stop": (C.c_int, [_ctxp, C.c_double]),
    "knp_emi_solve": (C.c_int, [_ctxp, C.c_double, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_int), _f64p]),
    "knp_knp_solve": (C.c_int, [_ctxp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _f64p]),
    "knp_set_knp_krylov": (C.c_int, [_ctxp, C.c_int, C.c_int]),
    "knp_set_emi_dg_smoother": (C.c_int, [_ctxp, C.c_int]),
    "knp_step_updates": (C.c_int, [_ctxp]),
    "knp_nernst": (C.c_int, [_ctxp]),
    "knp_picard_updates": (C.c_int, [_ctxp]),
    "knp_max_abs_diff": (C.c_int, [_ctxp, C.c_int, C.c_int, _f64p]),
    "knp_facet_trace": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "knp_sync": (C.c_int, [_ctxp]),
    "knp_timer_begin": (C.c_int, [_ctxp]),
    "knp_timer_end": (C.c_int, [_ctxp, C.POINTER(C.c_float)]),
    "knp_bench_apply": (C.c_int, [_ctxp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "knp_probe_facet_contraction": (C.c_int, [_ctxp, C.c_int, C.c_int64, C.c_int, _f64p, _f64p, C.POINTER(C.c_float)]),
    "knp_apply_timing": (C.c_int, [_ctxp, C.c_int]),
    "knp_apply_timing_read": (C.c_int, [_ctxp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "knp_apply_variant": (C.c_int, [_ctxp, C.c_int]),
    "knp_host_spgemm": (C.c_int, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _i32p, C.POINTER(_i32p),
                                  C.POINTER(_f64p), C.c_int]),
    "knp_host_spmv": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, _f64p, C.c_int]),
    "knp_host_free": (None, [C.c_void_p]),
    "knp_host_cell_gram": (C.c_int, [C.c_int64, C.c_int, _f64p, _i32p, _f64p, _f64p, C.c_int]),
    "knp_host_segment_sum": (C.c_int, [C.c_int64, _i64p, _i64p, _f64p, _f64p, C.c_int]),
    "knp_host_build_facets": (C.c_int64, [C.c_int64, C.c_int, _i32p, _i32p, _i32p, _i32p, _i8p]),
    "knp_host_geometry_classes": (C.c_int64, [C.c_int64, _f64p, _i32p, _i32p, _i8p, C.c_double, C.c_int64, _i32p, _i64p, C.c_int]),
    "knp_host_block_pattern": (C.c_int, [C.c_int64, C.c_int, C.c_int64, _i32p, _i64p, _i64p, _i32p, _i32p, _i64p, C.c_int]),
    "knp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "knp_comm_init": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_char_p]),
    "knp_comm_init_halo": (C.c_int, [_ctxp, C.c_char_p]),
    "knp_comm_init_shm": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_char_p, C.c_int64, C.c_int64]),
    "knp_set_interior": (C.c_int, [_ctxp, C.c_int64]),
    "knp_halo_tables": (C.c_int, [_ctxp, C.c_int, _i32p, _i64p, _i32p, _i64p, _i64p]),
    "knp_halo_exchange": (C.c_int, [_ctxp, C.c_int]),
    "knp_allreduce_sum": (C.c_int, [_ctxp, _f64p, C.c_int]),
    "knp_ode_create": (C.c_int, [_ctxp, C.c_int, C.c_int64, _i32p, C.c_int, C.c_int, _f64p, _f64p]),
    "knp_ode_table": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_int, _f64p]),
    "knp_ode_exchange": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]),
    "knp_ode_exchange_multi": (C.c_int, [_ctxp, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i64p, C.c_int]),
    "knp_ode_set_stimulus": (C.c_int, [_ctxp, C.c_int, C.c_int, _i32p, _f64p, C.POINTER(C.c_uint8)]),
    "knp_ode_step": (C.c_int, [_ctxp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "knp_amg_begin": (C.c_int, [_ctxp, C.c_int, C.c_int64, _i32p, _i32p, _i32p]),
    "knp_amg_columns": (C.c_int, [_ctxp, C.c_int, C.c_int]),
    "knp_amg_level": (C.c_int, [_ctxp, C.c_int, C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_double, C.c_int, C.c_double,
                                C.c_int64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p]),
    "knp_amg_finish": (C.c_int, [_ctxp, C.c_int, C.c_int64, _f64p]),
    "knp_amg_finish_f32": (C.c_int, [_ctxp, C.c_int, C.c_int64, _f32p]),
    "knp_host_sym_to_f32": (C.c_int, [C.c_int64, _f64p, _f32p, _f64p, C.c_int]),
    "knp_host_morton_order": (C.c_int, [C.c_int64, C.c_int, _f64p, _i32p, C.c_int, _f64p, _i64p, C.c_int]),
    "knp_host_cell_extent_median": (C.c_int, [C.c_int64, C.c_int, C.c_int, _f64p, _i32p, _f64p]),
    "knp_host_cell_neighbours": (C.c_int, [C.c_int64, C.c_int, _i32p, _i32p, _i8p, _i32p, _i8p, C.c_int]),
    "knp_host_box_marks": (C.c_int, [C.c_int64, C.c_int, _f64p, _i32p, C.c_int, _f64p, _f64p, C.c_double, C.c_int, C.POINTER(C.c_uint8), C.c_int]),
    "knp_host_strength": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, _f64p, C.c_double, _i32p, C.POINTER(_i32p), C.c_int]),
    "knp_host_smooth_prolongator": (C.c_int, [C.c_int64, C.c_int64, _i32p, _i32p, _f64p, _f64p, _i32p, _i32p, _f64p, _i32p, C.POINTER(_i32p),
                                            C.POINTER(_f64p), C.c_int]),
    "knp_host_truncate_prolongator": (C.c_int, [C.c_int64, _i32p, _i32p, _f64p, C.c_double, _f64p, _i32p, C.POINTER(_i32p), C.POINTER(_f64p), C.c_int]),
    "knp_host_mis2_aggregate": (C.c_int64, [C.c_int64, _i32p, _i32p, _f64p, _i64p, C.c_int]),
    "knp_amg_clear": (C.c_int, [_ctxp, C.c_int]),
    "knp_amg_interface": (C.c_int, [_ctxp, C.c_int64, C.c_int, _i32p, _i64p, _i32p, C.c_int64, _i32p, _i32p, _i32p]),
    "knp_amg_dist0": (C.c_int, [_ctxp, C.c_int]),
}

_lib = None


class KnpError(RuntimeError):
    pass


def load():
    """Load the shared library (no GPU needed for loading)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KnpError("libknpemi_hip.so is not built (%s); run `python __graft_entry__.py` or "
                           "knp-emi-dg_amd/build.py -- there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _p(a, typ):
    return a.ctypes.data_as(typ) if a is not None else None


def host_spgemm(A, B, nthreads=0):
    """C = A @ B for scipy CSR matrices through the library's threaded Gustavson product (sorted indices)."""
    import scipy.sparse as sp
    lib = load()
    A = A.tocsr(); B = B.tocsr()
    n, m = A.shape[0], B.shape[1]
    assert A.shape[1] == B.shape[0]
    Ap, Aj = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
    Bp, Bj = np.ascontiguousarray(B.indptr, dtype=np.int32), np.ascontiguousarray(B.indices, dtype=np.int32)
    Ax, Bx = np.ascontiguousarray(A.data, dtype=np.float64), np.ascontiguousarray(B.data, dtype=np.float64)
    Cp = np.empty(n + 1, dtype=np.int32)
    cj, cx = _i32p(), _f64p()
    rc = lib.knp_host_spgemm(n, m, _p(Ap, _i32p), _p(Aj, _i32p), _p(Ax, _f64p), _p(Bp, _i32p), _p(Bj, _i32p), _p(Bx, _f64p),
                             _p(Cp, _i32p), C.byref(cj), C.byref(cx), int(nthreads))
    if rc != 0:
        raise KnpError("knp_host_spgemm failed (%d)" % rc)
    nnz = int(Cp[-1])
    try:
        Cj = np.ctypeslib.as_array(cj, shape=(max(nnz, 1),))[:nnz].copy()
        Cx = np.ctypeslib.as_array(cx, shape=(max(nnz, 1),))[:nnz].copy()
    finally:
        lib.knp_host_free(cj)
        lib.knp_host_free(cx)
    out = sp.csr_matrix((Cx, Cj, Cp), shape=(n, m))
    out.has_sorted_indices = True
    return out


_T_STAMP = [None]


def _stamp(label):
    """KNP_DEBUG_SETUP=1: wall-clock stamps of the host setup stages on stderr (tools/profile_setup.py reads them off a bench run)."""
    if os.environ.get("KNP_DEBUG_SETUP", "0") != "1":
        return
    import sys
    import time
    now = time.perf_counter()
    if _T_STAMP[0] is None:
        _T_STAMP[0] = (now, now)
    t0, last = _T_STAMP[0]
    print("[knp setup %7.3f s  +%6.3f  pid %d  abs %.3f] %s" % (now - t0, now - last, os.getpid(), now, label), file=sys.stderr, flush=True)
    _T_STAMP[0] = (t0, now)


def _morton_native(pts, conn, scale):
    """morton_order through the library (csrc/host_sparse.cpp: knp_host_morton_order; the same order bit for bit): of the points, or of
    the midpoints of the rows of conn.  None if the library is not there."""
    try:
        lib = load()
    except OSError:
        return None
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n = pts.shape[0] if conn is None else conn.shape[0]
    order = np.empty(n, dtype=np.int64)
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float64)
    cn = None if conn is None else np.ascontiguousarray(conn, dtype=np.int32)
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    rc = lib.knp_host_morton_order(n, pts.shape[1], _p(pts, _f64p), _p(cn, _i32p), 0 if cn is None else cn.shape[1], _p(sc, _f64p),
                                   _p(order, _i64p), threads)
    return order if rc == 0 else None


def morton_order(points, scale=None):
    """Stable argsort of points along a Morton (Z-order) curve.  `scale` = typical cell extent per axis, so
    that the curve's bricks are cubes in units of CELLS (anisotropic meshes: 1 x 0.1 x 0.1 um boxes).
    Device cell numbering follows this curve: ~85 % of a cell's facet neighbours then sit in the same
    workgroup (served from LDS) and the rest is close enough to still be in the XCD's 4 MiB L2."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.ndim == 2 and pts.shape[0] >= 50000 and pts.shape[1] <= 3 and os.environ.get("KNP_SETUP_NATIVE_MORTON", "1") != "0":
        order = _morton_native(pts, None, scale)
        if order is not None:
            return order
    if scale is not None:
        pts = pts / np.asarray(scale, dtype=np.float64)
    n, d = pts.shape
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    lo = pts.min(axis=0)
    span = float((pts.max(axis=0) - lo).max())
    bits = 21 if d == 3 else 31
    q = np.zeros((n, d), dtype=np.uint64) if span == 0 else \
        np.minimum(((pts - lo) / span * (2 ** bits - 1)).astype(np.uint64), np.uint64(2 ** bits - 1))
    code = np.zeros(n, dtype=np.uint64)
    for b in range(bits):
        for k in range(d):
            code |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * d + k)
    return np.argsort(code, kind="stable")


def geometry_classes(mesh, order, max_classes=4096, tol=1.0e-9):
    """Group cells whose own shape and neighbour-apex positions coincide (relative tolerance `tol` of the
    cell size) and precompute one geometry record per group (layout: include/knpemi_hip.h,
    knp_set_geometry_classes).  Returns (cls uint16[nc] in device order, table [ncls, 36]) or None when the
    mesh is not (block-)structured enough (more than `max_classes` distinct shapes) or not 3D."""
    d = mesh.gdim
    if d != 3:
        return None
    nc, nv = mesh.cells.shape
    nb = nj = None
    if nc >= 20000:
        try:
            cf32 = np.ascontiguousarray(mesh.cell_facets, dtype=np.int32)
            fc32 = np.ascontiguousarray(mesh.facet_cells, dtype=np.int32)
            fl8 = np.ascontiguousarray(mesh.facet_local, dtype=np.int8)
            nb, nj = np.empty((nc, nv), dtype=np.int32), np.empty((nc, nv), dtype=np.int8)
            if load().knp_host_cell_neighbours(nc, nv, _p(cf32, _i32p), _p(fc32, _i32p), _p(fl8, _i8p), _p(nb, _i32p), _p(nj, _i8p), 0) != 0:
                nb = nj = None
        except OSError:
            nb = nj = None
    if nb is None:
        fc, fl = mesh.facet_cells, mesh.facet_local.astype(np.int64)
        cf = mesh.cell_facets                                   # [nc, 4] facet ids
        side = (fc[cf, 0] != np.arange(nc)[:, None]).astype(np.int64)          # which side of the facet this cell is
        nb = np.take_along_axis(fc[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]     # neighbour cell or -1
        nj = np.take_along_axis(fl[cf], (1 - side)[:, :, None], axis=2)[:, :, 0]     # neighbour's local facet
    _stamp("classes: neighbour table")
    native = _geometry_classes_native(mesh, order, nb, nj, max_classes, tol) if nc >= 20000 else None
    _stamp("classes: grouped")
    if native is not None:
        return native if native != "unstructured" else None
    X = mesh.coords[mesh.cells]                             # [nc, 4, 3]
    X0 = X[:, 0]
    has = nb >= 0
    apex_v = mesh.cells[np.maximum(nb, 0), np.maximum(nj, 0)]
    apex = np.where(has[:, :, None], mesh.coords[apex_v] - X0[:, None, :], 0.0)
    h2 = np.zeros(nc)
    for a in range(4):                                       # longest of the six edges (an [nc, 4, 4, 3] difference tensor was 0.4 s)
        for b in range(a + 1, 4):
            ed = X[:, a] - X[:, b]
            np.maximum(h2, np.einsum("cd,cd->c", ed, ed), out=h2)
    h = np.sqrt(h2)
    hN = np.where(has, h[np.maximum(nb, 0)], 0.0)
    feat = np.concatenate([(X[:, 1:] - X[:, :1]).reshape(nc, 9), apex.reshape(nc, 12), h[:, None], hN], axis=1)
    q = np.round(feat / (np.median(h) * tol)).astype(np.int64)
    rng = np.random.default_rng(12345)
    h1 = (q * rng.integers(1, 2 ** 62, size=q.shape[1], dtype=np.int64)[None, :]).sum(axis=1)     # wraps mod 2^64
    h2 = (q * rng.integers(1, 2 ** 62, size=q.shape[1], dtype=np.int64)[None, :]).sum(axis=1)
    _, first, inv = np.unique(h1, return_index=True, return_inverse=True)       # 1-D sort; the second hash and q itself are checked below
    inv = inv.ravel()
    ncls = len(first)
    if ncls > max_classes or ncls > max(64, nc // 8):
        return None          # not (block-)structured: the coordinate-path kernels are the right tool
    if not np.array_equal(h2[first][inv], h2) or not np.array_equal(q[first][inv], q):     # hash collision (astronomically unlikely): give up
        return None
    # records of the representatives
    Xr = X[first]
    J = (Xr[:, 1:, :] - Xr[:, :1, :]).transpose(0, 2, 1)
    Jinv = np.linalg.inv(J)
    g = np.empty((ncls, 4, 3))
    g[:, 1:, :] = Jinv
    g[:, 0, :] = -Jinv.sum(axis=1)
    G = np.einsum("cad,cbd->cab", g, g)
    vol = np.abs(np.linalg.det(J)) / 6.0
    table = np.zeros((ncls, 36))
    table[:, 0] = vol
    k = 1
    for a in range(4):
        for b in range(a, 4):
            table[:, k] = G[:, a, b]
            k += 1
    L = np.einsum("cad,cid->cia", g, apex[first])          # [cls, facet i, vertex a]
    L[:, :, 0] += 1.0
    hasr = has[first]
    for i in range(4):
        table[:, 11 + 6 * i:11 + 6 * i + 4] = np.where(hasr[:, i:i + 1], L[:, i, :], 0.0)
        table[:, 11 + 6 * i + 4] = np.sqrt(G[:, i, i])
        table[:, 11 + 6 * i + 5] = np.where(hasr[:, i], 2.0 / (h[first] + np.where(hasr[:, i], hN[first, i], 1.0)), 0.0)
    return np.ascontiguousarray(inv[order].astype(np.uint16)), np.ascontiguousarray(table)


def _class_table(mesh, first, nb, nj):
    """Geometry records [ncls, 36] of the class representatives `first` (layout: include/knpemi_hip.h, knp_set_geometry_classes)."""
    ncls = len(first)
    X = mesh.coords[mesh.cells[first]]
    nbf, njf = nb[first], nj[first]
    has = nbf >= 0
    apex = np.where(has[:, :, None], mesh.coords[mesh.cells[np.maximum(nbf, 0), np.maximum(njf, 0)]] - X[:, :1, :], 0.0)

    def diam(Xc):
        e = Xc[:, :, None, :] - Xc[:, None, :, :]
        return np.sqrt((e ** 2).sum(axis=3).max(axis=(1, 2)))
    h = diam(X)
    hN = np.where(has, diam(mesh.coords[mesh.cells[np.maximum(nbf, 0).ravel()]]).reshape(ncls, 4), 0.0)
    J = (X[:, 1:, :] - X[:, :1, :]).transpose(0, 2, 1)
    Jinv = np.linalg.inv(J)
    g = np.empty((ncls, 4, 3))
    g[:, 1:, :] = Jinv
    g[:, 0, :] = -Jinv.sum(axis=1)
    G = np.einsum("cad,cbd->cab", g, g)
    table = np.zeros((ncls, 36))
    table[:, 0] = np.abs(np.linalg.det(J)) / 6.0
    k = 1
    for a in range(4):
        for b in range(a, 4):
            table[:, k] = G[:, a, b]
            k += 1
    L = np.einsum("cad,cid->cia", g, apex)
    L[:, :, 0] += 1.0
    for i in range(4):
        table[:, 11 + 6 * i:11 + 6 * i + 4] = np.where(has[:, i:i + 1], L[:, i, :], 0.0)
        table[:, 11 + 6 * i + 4] = np.sqrt(G[:, i, i])
        table[:, 11 + 6 * i + 5] = np.where(has[:, i], 2.0 / (h + np.where(has[:, i], hN[:, i], 1.0)), 0.0)
    return table


def _geometry_classes_native(mesh, order, nb, nj, max_classes, tol):
    """geometry_classes through the library's threaded hash grouping (csrc/host_sparse.cpp: knp_host_geometry_classes): the same
    26 quantised features per cell, classes numbered by first appearance.  None: library not available (numpy path)."""
    try:
        lib = load()
    except Exception:
        return None
    nc = mesh.cells.shape[0]
    X = mesh.coords[mesh.cells[::max(1, nc // 4096)]]
    e = X[:, :, None, :] - X[:, None, :, :]
    hmed = float(np.median(np.sqrt((e ** 2).sum(axis=3).max(axis=(1, 2)))))          # scale of the tolerance (a sample of the cells)
    coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
    cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
    nb32 = np.ascontiguousarray(nb, dtype=np.int32)
    nj8 = np.ascontiguousarray(nj, dtype=np.int8)
    limit = int(min(max_classes, max(64, nc // 8)))
    cls = np.empty(nc, dtype=np.int32)
    first = np.empty(limit, dtype=np.int64)
    n = int(lib.knp_host_geometry_classes(nc, _p(coords, _f64p), _p(cells, _i32p), _p(nb32, _i32p), _p(nj8, _i8p), hmed * tol, limit,
                                          _p(cls, _i32p), _p(first, _i64p), 0))
    if n == -3 or n == -4:
        return "unstructured"
    if n < 0:
        return None
    first = first[:n]
    return np.ascontiguousarray(cls[order].astype(np.uint16)), np.ascontiguousarray(_class_table(mesh, first, nb, nj))


class Device:
    """One context = one GPU = one partition of the mesh."""

    def __init__(self, mesh, cell_tags, facet_tags, membrane_tags, n_ions, degree=1, device=0, nc_owned=None,
                 reorder=None):
        self.lib = load()
        self.ctx = _ctxp()
        _stamp("device: start")
        nc = mesh.cells.shape[0]
        n_own = nc if nc_owned is None else int(nc_owned)
        if reorder is None:
            reorder = os.environ.get("KNP_NO_REORDER", "0") != "1"
        # device cell numbering: owned cells along a Morton curve, ghosts keep their (peer-grouped) places
        order = np.arange(nc, dtype=np.int64)
        scale = None
        self.n_interior = n_own
        if reorder and nc:
            native = nc >= 50000 and mesh.gdim <= 3 and os.environ.get("KNP_SETUP_NATIVE_MORTON", "1") != "0"
            co = o = None
            if native:
                # median cell extent and the curve over the cell midpoints in the library (same numbers, same order): 0.43 -> 0.05 s at 10^6 tets
                co = np.ascontiguousarray(mesh.coords, dtype=np.float64)
                cl = np.ascontiguousarray(mesh.cells, dtype=np.int32)
                scale = np.empty(mesh.gdim)
                if self.lib.knp_host_cell_extent_median(nc, cl.shape[1], mesh.gdim, _p(co, _f64p), _p(cl, _i32p), _p(scale, _f64p)) == 0:
                    scale = np.maximum(scale, 1e-300)
                    o = _morton_native(co, cl[:n_own], scale)
            if o is None:
                xc = mesh.coords[mesh.cells]
                scale = np.maximum(np.median(xc.max(axis=1) - xc.min(axis=1), axis=0), 1e-300)
                o = morton_order(mesh.cell_midpoints()[:n_own], scale)
            order[:n_own] = o
            if n_own < nc:
                # a partition: owned cells without a ghost neighbour first (their part of an apply runs while the halo
                # exchange is in flight), the cells on the cut after them; both groups along the Morton curve
                fcs = np.asarray(mesh.facet_cells)
                cut = fcs[(fcs[:, 1] >= 0) & ((fcs[:, 0] >= n_own) != (fcs[:, 1] >= n_own))]
                on_cut = np.zeros(nc, dtype=bool)
                on_cut[cut.ravel()] = True
                o = order[:n_own]
                order[:n_own] = np.concatenate([o[~on_cut[o]], o[on_cut[o]]])
                self.n_interior = int((~on_cut[:n_own]).sum())
        rank = np.empty(nc, dtype=np.int64)
        rank[order] = np.arange(nc)
        self.cell_order, self.cell_rank = order, rank            # device -> caller, caller -> device
        _stamp("device: cell order (Morton)")
        assert (np.diff(mesh.cells, axis=1) > 0).all(), "cells must hold ascending vertex ids"
        # vertex STORAGE order (ids are only used to fetch coordinates on the device; the local vertex order
        # inside each cell -- which carries the facet matching -- is untouched)
        nv = mesh.coords.shape[0]
        vorder = morton_order(mesh.coords, scale) if reorder else np.arange(nv, dtype=np.int64)
        vrank = np.empty(nv, dtype=np.int64)
        vrank[vorder] = np.arange(nv)
        self.vertex_rank = vrank                                  # caller vertex id -> storage id
        _stamp("device: vertex order")
        coords = np.ascontiguousarray(mesh.coords[vorder], dtype=np.float64)
        cells = np.ascontiguousarray(vrank[mesh.cells[order]], dtype=np.int32)
        ctags = np.ascontiguousarray(np.asarray(cell_tags)[order], dtype=np.uint32)
        ftags = np.ascontiguousarray(np.asarray(facet_tags), dtype=np.uint32)
        fc = np.asarray(mesh.facet_cells, dtype=np.int64)
        fcells = np.ascontiguousarray(np.where(fc >= 0, rank[np.maximum(fc, 0)], -1), dtype=np.int32)
        flocal = np.ascontiguousarray(mesh.facet_local, dtype=np.int8)
        mt = np.ascontiguousarray(np.asarray(list(membrane_tags)), dtype=np.uint32)
        assert ctags.shape == (nc,) and ftags.shape == (fcells.shape[0],)
        _stamp("device: renumbered tables")
        self.dim = mesh.gdim
        self.nd = self.dim + 1 if degree == 1 else (self.dim + 1) * (self.dim + 2) // 2
        self.nc = nc
        self.nc_owned = nc if nc_owned is None else int(nc_owned)
        self.nf = fcells.shape[0]
        self.n_ions = int(n_ions)
        self.n_sys = self.n_ions - 1
        rc = self.lib.knp_ctx_create(C.byref(self.ctx), device, self.dim, degree, self.n_ions, coords.shape[0], nc,
                                     self.nc_owned, self.nf, _p(coords, _f64p), _p(cells, _i32p), _p(ctags, _u32p),
                                     _p(fcells, _i32p), _p(flocal, _i8p), _p(ftags, _u32p), len(mt), _p(mt, _u32p))
        if rc != 0:
            msg = self.lib.knp_last_error(None)
            self.ctx = None
            raise KnpError("knp_ctx_create failed (%d): %s" % (rc, msg.decode() if msg else "?"))
        if self.n_interior != n_own:
            self._chk(self.lib.knp_set_interior(self.ctx, self.n_interior), "knp_set_interior")
        self.n_geometry_classes = 0
        _stamp("device: knp_ctx_create")
        self.nranks = 1
        self.degree = int(degree)
        self._pending = []                 # queued PDE<->ODE column copies: (handle, to_facet, what, col, field, offset)
        self._tmp_slot = -1                # rotating scratch slot of facet_trace results
        self._tmp_gen = [0] * FACET_TMP_SLOTS
        if degree != 1:
            # DG-p path: the device integrates the forms with host-tabulated rules (csrc/tab_dg.hip)
            from knpemidg import dgtab
            for slot, (nloc, nq, w, B, dB) in dgtab.tables(self.dim, degree).items():
                self._chk(self.lib.knp_set_tabulation(self.ctx, slot, nloc, nq, _p(w, _f64p), _p(B, _f64p), _p(dB, _f64p)),
                          "knp_set_tabulation")
        if os.environ.get("KNP_NO_CLASSES", "0") != "1":
            gc = geometry_classes(mesh, order)
            if gc is not None:
                cls, table = gc
                self._chk(self.lib.knp_set_geometry_classes(self.ctx, table.shape[0], _p(cls, C.POINTER(C.c_uint16)),
                                                            _p(table, _f64p)), "knp_set_geometry_classes")
                self.n_geometry_classes = table.shape[0]
        _stamp("device: geometry classes set")

    # -- helpers -------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc != 0:
            msg = self.lib.knp_last_error(self.ctx)
            raise KnpError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.knp_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, C_M, dt, F, R, T, C_phi, tau_emi, tau_knp, z, D, rho=None, fsrc=None, splitting=True):
        z = np.ascontiguousarray(z, dtype=np.float64)
        D = np.asarray(D, dtype=np.float64)
        assert z.shape == (self.n_ions,) and D.shape == (self.n_ions, self.nc)
        o = self.cell_order
        D = np.ascontiguousarray(D[:, o])
        rho = None if rho is None else np.ascontiguousarray(np.asarray(rho, dtype=np.float64)[o])
        fsrc = None if fsrc is None else np.ascontiguousarray(np.asarray(fsrc, dtype=np.float64)[:, o])
        self._chk(self.lib.knp_set_params(self.ctx, C_M, dt, F, R, T, C_phi, tau_emi, tau_knp, _p(z, _f64p), _p(D, _f64p),
                                          _p(rho, _f64p), _p(fsrc, _f64p), int(splitting)), "knp_set_params")

    def set_mms(self, C, extra_emi, extra_knp):
        """Manufactured-solution data (caller cell order): C [n_sys, nc], extra_emi [nc, nd], extra_knp [n_sys, nc, nd]."""
        o = self.cell_order
        C_ = np.ascontiguousarray(np.asarray(C, dtype=np.float64)[:, o])
        ee = np.ascontiguousarray(np.asarray(extra_emi, dtype=np.float64).reshape(self.nc, self.nd)[o])
        ek = np.ascontiguousarray(np.asarray(extra_knp, dtype=np.float64).reshape(self.n_sys, self.nc, self.nd)[:, o])
        self._chk(self.lib.knp_set_mms(self.ctx, _p(C_, _f64p), _p(ee, _f64p), _p(ek, _f64p)), "knp_set_mms")

    def set_source(self, src):
        """Load vector of non-constant ion sources, [n_sys, nc, nd] in caller cell order (None clears)."""
        if src is not None:
            src = np.ascontiguousarray(np.asarray(src, dtype=np.float64).reshape(self.n_sys, self.nc, self.nd)[:, self.cell_order])
        self._chk(self.lib.knp_set_source(self.ctx, _p(src, _f64p)), "knp_set_source")

    def size(self, field):
        return int(self.lib.knp_field_size(self.ctx, field))

    _FACET_FIELDS = (F_PHI_M, F_I_CH, F_E, F_FACET_TMP)

    def _nodal_blocks(self, field, offset, count):
        """Nodal fields are stored in device cell order; transfers address whole [nc*nd] blocks."""
        ndof = self.nc * self.nd
        if field in self._FACET_FIELDS:
            return None
        if offset % ndof or count % ndof:
            raise KnpError("nodal field transfers must cover whole [nc*nd] blocks")
        return count // ndof

    def upload(self, field, a, offset=0):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        nb = self._nodal_blocks(field, offset, a.size)
        if nb is not None:
            a = np.ascontiguousarray(a.reshape(nb, self.nc, self.nd)[:, self.cell_order]).ravel()
        self._chk(self.lib.knp_upload(self.ctx, field, _p(a, _f64p), offset, a.size), "knp_upload")

    def download(self, field, offset=0, count=None):
        n = self.size(field) - offset if count is None else count
        out = np.empty(n, dtype=np.float64)
        self._chk(self.lib.knp_download(self.ctx, field, _p(out, _f64p), offset, n), "knp_download")
        nb = self._nodal_blocks(field, offset, n)
        if nb is not None:
            out = np.ascontiguousarray(out.reshape(nb, self.nc, self.nd)[:, self.cell_rank]).ravel()
        return out

    _DT_DTYPE = {0: np.int32, 1: np.int32, 2: np.uint32, 3: np.int32, 4: np.int32, 5: np.int32, 6: np.uint16, 7: np.int64}

    def debug_table(self, which):
        """Connectivity table `which` (include/knpemi_hip.h: knp_debug_table_id) as the library derived it, flat, in DEVICE cell
        order (device cell d = caller cell `cell_order[d]`)."""
        n = int(self.lib.knp_debug_table_size(self.ctx, which))
        if n < 0:
            raise KnpError("unknown debug table %d" % which)
        out = np.zeros(n // np.dtype(self._DT_DTYPE[which]).itemsize, dtype=self._DT_DTYPE[which])
        self._chk(self.lib.knp_debug_table(self.ctx, which, out.ctypes.data_as(C.c_void_p), n), "knp_debug_table")
        return out

    def copy_field(self, dst, src):
        self._chk(self.lib.knp_copy_field(self.ctx, dst, src), "knp_copy_field")

    def update_kappa(self):
        self._chk(self.lib.knp_update_kappa(self.ctx), "knp_update_kappa")

    def update_dnphi(self):
        self._chk(self.lib.knp_update_dnphi(self.ctx), "knp_update_dnphi")

    def emi_apply(self, fx=F_X, fy=F_Y):
        self._chk(self.lib.knp_emi_apply(self.ctx, fx, fy), "knp_emi_apply")

    def knp_apply(self, fx=F_X, fy=F_Y):
        self._chk(self.lib.knp_knp_apply(self.ctx, fx, fy), "knp_knp_apply")

    def emi_rhs(self):
        self._chk(self.lib.knp_emi_rhs(self.ctx), "knp_emi_rhs")

    def knp_rhs(self):
        self._chk(self.lib.knp_knp_rhs(self.ctx), "knp_knp_rhs")

    def emi_residual_target(self, r_abs):
        """r_abs > 0: the following EMI solves stop on ||b - A phi||_w <= r_abs (include/knpemi_hip.h); 0: preconditioned-norm test."""
        self._chk(self.lib.knp_emi_residual_target(self.ctx, float(r_abs)), "knp_emi_residual_target")

    def emi_solve(self, rtol, atol=1e-40, maxit=1000, check_every=25):
        it = C.c_int(0)
        res = np.zeros(3)
        rc = self.lib.knp_emi_solve(self.ctx, rtol, atol, maxit, check_every, C.byref(it), _p(res, _f64p))
        self._chk(rc, "knp_emi_solve")
        return it.value, res

    def knp_solve(self, rtol, atol=1e-40, maxit=1000, min_it=5, check_every=10):
        it = (C.c_int * max(self.n_sys, 1))()
        res = np.zeros(3 * self.n_sys)
        rc = self.lib.knp_knp_solve(self.ctx, rtol, atol, maxit, min_it, check_every, it, _p(res, _f64p))
        self._chk(rc, "knp_knp_solve")
        return list(it)[:self.n_sys], res.reshape(self.n_sys, 3)

    def set_knp_krylov(self, method, restart=30):
        """KNP Krylov method: 'bicgstab' (default) or 'gmres' (restarted, the reference's GMRES(30): solver.py:684-701)."""
        code = {"bicgstab": 0, "gmres": 1}[method] if isinstance(method, str) else int(method)
        self._chk(self.lib.knp_set_knp_krylov(self.ctx, code, int(restart)), "knp_set_knp_krylov")

    def set_emi_dg_smoother(self, chebyshev):
        """DG-level smoother of the EMI preconditioner: True = two-step Chebyshev block-Jacobi, False = plain block-Jacobi, None = default."""
        self._chk(self.lib.knp_set_emi_dg_smoother(self.ctx, -1 if chebyshev is None else int(bool(chebyshev))), "knp_set_emi_dg_smoother")

    def step_updates(self):
        self._chk(self.lib.knp_step_updates(self.ctx), "knp_step_updates")

    def picard_updates(self):
        self._chk(self.lib.knp_picard_updates(self.ctx), "knp_picard_updates")

    def max_abs_diff(self, fa, fb):
        out = np.zeros(1)
        self._chk(self.lib.knp_max_abs_diff(self.ctx, fa, fb, _p(out, _f64p)), "knp_max_abs_diff")
        return float(out[0])

    def nernst(self):
        self._chk(self.lib.knp_nernst(self.ctx), "knp_nernst")

    def facet_trace(self, field, species, side, download=True):
        """Facet average of a trace into the next scratch slot of F_FACET_TMP.  download=True returns the values
        ([nf] array); otherwise returns (slot, generation) -- the slot is recycled after FACET_TMP_SLOTS further
        projections, `tmp_slot_valid` tells whether a result is still there."""
        self._tmp_slot = slot = (self._tmp_slot + 1) % FACET_TMP_SLOTS
        # queued ODE copies that still read this slot must run before it is overwritten
        if any(f == F_FACET_TMP and off == slot * self.nf for (_, _, _, _, f, off) in self._pending):
            self._flush()
        self._tmp_gen[slot] += 1
        self._chk(self.lib.knp_facet_trace(self.ctx, field, species, side, slot), "knp_facet_trace")
        if download:
            return self.download(F_FACET_TMP, slot * self.nf, self.nf)
        return slot, self._tmp_gen[slot]

    def tmp_slot_valid(self, slot, gen):
        return self._tmp_gen[slot] == gen

    def sync(self):
        self._chk(self.lib.knp_sync(self.ctx), "knp_sync")

    def timer_begin(self):
        self._chk(self.lib.knp_timer_begin(self.ctx), "knp_timer_begin")

    def timer_end(self):
        ms = C.c_float(0)
        self._chk(self.lib.knp_timer_end(self.ctx, C.byref(ms)), "knp_timer_end")
        return ms.value

    def bench_apply(self, which, reps):
        ms = C.c_float(0)
        self._chk(self.lib.knp_bench_apply(self.ctx, which, reps, C.byref(ms)), "knp_bench_apply")
        return ms.value

    def apply_timing(self, enable):
        self._chk(self.lib.knp_apply_timing(self.ctx, int(bool(enable))), "knp_apply_timing")

    def apply_variant(self, which):
        """Kernel family the operator apply dispatches to (include/knpemi_hip.h, knp_apply_variant)."""
        return int(self.lib.knp_apply_variant(self.ctx, which))

    def apply_timing_read(self, which):
        """(average ms, launches) of the operator applies issued inside the solves since the last read."""
        ms, n = C.c_float(0), C.c_int(0)
        self._chk(self.lib.knp_apply_timing_read(self.ctx, which, C.byref(ms), C.byref(n)), "knp_apply_timing_read")
        return ms.value, n.value

    def probe_facet_contraction(self, variant, inputs, reps=20):
        """Diagnostic: the P2 facet-quadrature contraction as FMA chain (0) or FP64 MFMA tiles (1); returns (out, ms)."""
        a = np.ascontiguousarray(inputs, dtype=np.float64)
        assert a.ndim == 2 and a.shape[1] == 26
        out = np.empty((a.shape[0], 9))
        ms = C.c_float(0)
        self._chk(self.lib.knp_probe_facet_contraction(self.ctx, int(variant), a.shape[0], int(reps), _p(a, _f64p), _p(out, _f64p),
                                                       C.byref(ms)), "knp_probe_facet_contraction")
        return out, ms.value

    # -- multi-GPU ---------------------------------------------------------------------
    def comm_init(self, rank, nranks, uid, uid_halo=None):
        self._chk(self.lib.knp_comm_init(self.ctx, rank, nranks, uid), "knp_comm_init")
        self.nranks = int(nranks)
        if uid_halo is not None:
            self._chk(self.lib.knp_comm_init_halo(self.ctx, uid_halo), "knp_comm_init_halo")

    def comm_init_shm(self, rank, nranks, name, red_doubles, out_doubles):
        """Host-staged shared-memory communicator (include/knpemi_hip.h: knp_comm_init_shm): validation runs with several ranks
        on one GPU."""
        self._chk(self.lib.knp_comm_init_shm(self.ctx, rank, nranks, name.encode(), int(red_doubles), int(out_doubles)), "knp_comm_init_shm")
        self.nranks = int(nranks)

    def set_interior(self, n_interior):
        self._chk(self.lib.knp_set_interior(self.ctx, int(n_interior)), "knp_set_interior")

    def halo_tables(self, peers, send_lists, recv_offsets, recv_counts):
        peers = np.ascontiguousarray(peers, dtype=np.int32)
        sc = np.ascontiguousarray([len(s) for s in send_lists], dtype=np.int64)
        cells = np.concatenate(send_lists) if len(send_lists) else np.zeros(0, dtype=np.int64)
        cells = np.ascontiguousarray(self.cell_rank[np.asarray(cells, dtype=np.int64)], dtype=np.int32)
        ro = np.ascontiguousarray(recv_offsets, dtype=np.int64)
        rcnt = np.ascontiguousarray(recv_counts, dtype=np.int64)
        self._chk(self.lib.knp_halo_tables(self.ctx, len(peers), _p(peers, _i32p), _p(sc, _i64p), _p(cells, _i32p),
                                           _p(ro, _i64p), _p(rcnt, _i64p)), "knp_halo_tables")

    # -- membrane ODEs on the device ------------------------------------------------------
    def ode_create(self, model, facets, states, params):
        facets = np.ascontiguousarray(facets, dtype=np.int32)
        states = np.ascontiguousarray(states, dtype=np.float64)
        params = np.ascontiguousarray(params, dtype=np.float64)
        h = self.lib.knp_ode_create(self.ctx, int(model), len(facets), _p(facets, _i32p), states.shape[1], params.shape[1],
                                    _p(states, _f64p), _p(params, _f64p))
        if h < 0:
            self._chk(h, "knp_ode_create")
        return h

    def ode_table(self, handle, what, shape, upload=None):
        a = np.empty(shape, dtype=np.float64) if upload is None else np.ascontiguousarray(upload, dtype=np.float64)
        self._chk(self.lib.knp_ode_table(self.ctx, handle, what, 0 if upload is None else 1, _p(a, _f64p)), "knp_ode_table")
        return a

    def ode_exchange(self, handle, what, col, field, row, to_facet):
        """Queue one PDE<->ODE column copy.  Consecutive copies of one membrane model and direction leave as ONE
        kernel (knp_ode_exchange_multi) the next time anything else touches the device (`_flush`)."""
        self._pending.append((int(handle), int(to_facet), int(what), int(col), int(field), int(row) * self.nf))

    def _flush(self):
        if not self._pending:
            return
        ops, self._pending = self._pending, []
        i = 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j][:2] == ops[i][:2]:
                j += 1
            grp = ops[i:j]
            what = np.ascontiguousarray([g[2] for g in grp], dtype=np.int32)
            col = np.ascontiguousarray([g[3] for g in grp], dtype=np.int32)
            fld = np.ascontiguousarray([g[4] for g in grp], dtype=np.int32)
            off = np.ascontiguousarray([g[5] for g in grp], dtype=np.int64)
            self._chk(self.lib.knp_ode_exchange_multi(self.ctx, grp[0][0], len(grp), _p(what, _i32p), _p(col, _i32p),
                                                      _p(fld, _i32p), _p(off, _i64p), grp[0][1]), "knp_ode_exchange_multi")
            i = j

    def ode_set_stimulus(self, handle, cols, values, mask):
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        values = np.ascontiguousarray(values, dtype=np.float64)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        self._chk(self.lib.knp_ode_set_stimulus(self.ctx, handle, len(cols), _p(cols, _i32p), _p(values, _f64p),
                                                _p(mask, C.POINTER(C.c_uint8))), "knp_ode_set_stimulus")

    def ode_step(self, handle, t0, dt, rtol=1.0e-8, atol=0.0):
        """Asynchronous (no host round trip); a failed node surfaces as KnpError at the next solve / sync / table read."""
        self._chk(self.lib.knp_ode_step(self.ctx, handle, t0, dt, rtol, atol), "knp_ode_step")

    # -- auxiliary-space AMG (knpemidg/amg.py builds, csrc/amg.hip applies) ------------------
    def amg_interface(self, n_local, peers, lists, uvtx, aptr, asrc):
        """Shared-dof tables of the row-distributed conforming level (amg.Dist0Space.interface_tables)."""
        peers = np.ascontiguousarray(peers, dtype=np.int32)
        cnt = np.ascontiguousarray([len(l) for l in lists], dtype=np.int64)
        idx = np.ascontiguousarray(np.concatenate(lists) if len(lists) else np.zeros(0), dtype=np.int32)
        uvtx = np.ascontiguousarray(uvtx, dtype=np.int32)
        aptr = np.ascontiguousarray(aptr, dtype=np.int32)
        asrc = np.ascontiguousarray(asrc if len(asrc) else np.zeros(1), dtype=np.int32)
        self._chk(self.lib.knp_amg_interface(self.ctx, int(n_local), len(peers), _p(peers, _i32p), _p(cnt, _i64p), _p(idx, _i32p), len(uvtx),
                                             _p(uvtx, _i32p), _p(aptr, _i32p), _p(asrc, _i32p)), "knp_amg_interface")

    def amg_upload(self, which, dg2cg, levels, ncol=1, dist0=False):
        """dg2cg [nc, nd]: conforming dof of every DG dof (caller's cell order); levels from amg.build_hierarchy.
        dist0: level 0 (and dg2cg) are this rank's rows in local numbering (amg.Dist0Space.localize; amg_interface first)."""
        nd = self.nd
        d2c = np.ascontiguousarray(np.asarray(dg2cg)[self.cell_order].ravel(), dtype=np.int32)
        assert d2c.shape == (self.nc * nd,)
        ncg = levels[0].A.shape[0]
        # (the inverse map conforming dof -> owned DG dofs is derived by the library: null pointers)
        self._chk(self.lib.knp_amg_begin(self.ctx, which, ncg, _p(d2c, _i32p), None, None), "knp_amg_begin")
        if ncol != 1:
            self._chk(self.lib.knp_amg_columns(self.ctx, which, int(ncol)), "knp_amg_columns")
        if dist0:
            self._chk(self.lib.knp_amg_dist0(self.ctx, which), "knp_amg_dist0")

        def csr(M):
            M = M.tocsr()
            M.sort_indices()
            return (np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                    np.ascontiguousarray(M.data, dtype=np.float64))
        for l, lv in enumerate(levels):
            rpA, ciA, vA = csr(lv.A)
            dinv = np.ascontiguousarray(lv.dinv, dtype=np.float64)
            last = l == len(levels) - 1
            if last:
                z32, z64 = np.zeros(1, np.int32), np.zeros(1, np.float64)
                args = (0, _p(z32, _i32p), _p(z32, _i32p), _p(z64, _f64p), _p(z32, _i32p), _p(z32, _i32p), _p(z64, _f64p))
            else:
                rpP, ciP, vP = csr(lv.P)
                rpR, ciR, vR = csr(lv.R)
                args = (lv.P.shape[1], _p(rpP, _i32p), _p(ciP, _i32p), _p(vP, _f64p), _p(rpR, _i32p), _p(ciR, _i32p), _p(vR, _f64p))
            self._chk(self.lib.knp_amg_level(self.ctx, which, lv.A.shape[0], _p(rpA, _i32p), _p(ciA, _i32p), _p(vA, _f64p),
                                             _p(dinv, _f64p), float(lv.rho), int(lv.cheb_degree), float(lv.cheb_lower), *args),
                      "knp_amg_level")
        pinv = levels[-1].pinv
        if pinv.dtype == np.float32:                               # already rounded by the setup (amg._coarse_pseudo_inverse)
            pinv = np.ascontiguousarray(pinv)
            self._chk(self.lib.knp_amg_finish_f32(self.ctx, which, pinv.shape[0], _p(pinv, _f32p)), "knp_amg_finish_f32")
        else:
            pinv = np.ascontiguousarray(pinv, dtype=np.float64)
            self._chk(self.lib.knp_amg_finish(self.ctx, which, pinv.shape[0], _p(pinv, _f64p)), "knp_amg_finish")

    def amg_clear(self, which):
        self._chk(self.lib.knp_amg_clear(self.ctx, which), "knp_amg_clear")

    def allreduce_sum(self, values):
        """Sum of a few host scalars over the ranks of a partitioned run (identical bits on every rank); the values themselves
        without a communicator."""
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
        self._chk(self.lib.knp_allreduce_sum(self.ctx, _p(v, _f64p), len(v)), "knp_allreduce_sum")
        return v

    def knp_early_stop(self, factor):
        self._chk(self.lib.knp_knp_early_stop(self.ctx, float(factor)), "knp_knp_early_stop")

    def knp_load_measure(self):
        """Per solved species: sum over the owned cells of (|b_K| / vol_K)^8 of the current KNP right-hand side (knp_knp_load_measure)."""
        out = np.zeros(self.n_sys)
        self._chk(self.lib.knp_knp_load_measure(self.ctx, _p(out, _f64p)), "knp_knp_load_measure")
        return out

    def halo_exchange(self, field):
        self._chk(self.lib.knp_halo_exchange(self.ctx, field), "knp_halo_exchange")


def _flushing(fn):
    def wrapped(self, *a, **k):
        if self._pending:
            self._flush()
        return fn(self, *a, **k)
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


# every call that reads or writes device data first issues the queued PDE<->ODE copies (order is preserved)
for _name in ("close", "set_params", "set_mms", "upload", "download", "copy_field", "update_kappa", "update_dnphi", "emi_apply",
              "knp_apply", "emi_rhs", "knp_rhs", "emi_solve", "knp_solve", "step_updates", "picard_updates", "max_abs_diff",
              "nernst", "sync", "timer_begin", "timer_end", "bench_apply", "ode_table", "ode_step", "ode_set_stimulus",
              "amg_upload", "amg_interface", "halo_exchange", "knp_load_measure", "apply_timing_read", "comm_init", "set_interior"):
    setattr(Device, _name, _flushing(getattr(Device, _name)))


def comm_unique_id():
    buf = C.create_string_buffer(128)
    if load().knp_comm_unique_id(buf) != 0:
        raise KnpError("knp_comm_unique_id failed")
    return buf.raw
