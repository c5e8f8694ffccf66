#!/usr/bin/env python3
"""Benchmark of the DG assemble-and-solve hot path on MI355X.

metric  : DoF-updates/sec per PDE timestep (BASELINE.json) = (phi DoFs + KNP DoFs) / wall-clock of one
          splitting step (ODE step + assemble-equivalent + EMI solve + KNP solve + step III).
workload: 3D idealized 4-axon mesh, refinement r (default 2: 995 328 tets, 11 943 936 P1 DoFs), Na/K/Cl +
          potential, HH membranes with stimulus -- BASELINE configs[3]'s mesh on N GPUs (slab partition
          in x, strong scaling: the total mesh is fixed as N grows).
Launch  : python bench.py --gpus 1 ...        or, for N > 1,
          python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
                 bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "idealized_geometries")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0                      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EMI_BYTES_PER_CELL = 137.0                 # algorithmic bytes / cell, 3D P1
KNP_BYTES_PER_CELL = 217.0                 # 2 species batched


# ---- cpu_baseline: the oracle's assembled-CSR path on the host cores ---------------------------------------------------------------
_CPU_PART = {}


def _cpu_part_init(resolution, world):
    """Worker state of the parallel assembly: the x-slab part `rank` of the mesh (owned cells + one ghost layer, exactly what a rank
    of the partitioned GPU run holds: knpemidg/partition.py) with the oracle problem on it.  Rows of owned cells assembled on the
    part are the rows of the global matrix (DG couples a cell to its facet neighbours only; tests/test_gpu_parity.py:
    test_partitioned_kernels_on_one_gpu checks the same statement for the kernels)."""
    for q in (os.path.join(ROOT, "oracle"), os.path.join(ROOT, "knp-emi-dg_amd")):
        if q not in sys.path:
            sys.path.insert(0, q)
    _CPU_PART.clear()
    _CPU_PART.update(resolution=resolution, world=world, parts={})


def _cpu_part_problem(rank):
    import knpemi_oracle as ko
    from knpemidg.mesh import make_mesh_3D
    from knpemidg.partition import Partition
    st = _CPU_PART
    if rank not in st["parts"]:
        if "mesh" not in st:
            st["mesh"] = make_mesh_3D(st["resolution"])
            st["partition"] = Partition(st["mesh"][0], st["world"], method="slab")
        m, s, f = st["mesh"]
        loc = st["partition"].local(rank)
        sub_l, surf_l = loc.localize(s, f, (1, 2))
        pb = ko.build_idealized(loc.mesh, sub_l.array(), surf_l.array())
        st["parts"][rank] = (loc, pb)
    return st["parts"][rank]


def _cpu_part_warm(rank):
    _cpu_part_problem(rank)
    return rank


def _cpu_part_assemble(job):
    """One part of one operator: kind 'emi' -> (rows of a_emi, L_emi), 'knp' -> (rows of A_knp,k, L_knp,k) of the owned cells, global
    column ids.  state: the global fields the forms read, sliced here to the part."""
    import knpemi_oracle as ko
    import scipy.sparse as sp
    kind, rank, k, state = job
    loc, pb = _cpu_part_problem(rank)
    cg, fg, no, nd = loc.cells_global, loc.facets_global, loc.nc_owned, pb.nd
    pb.phi_M = state["phi_M"][fg]
    for name in pb.I_ch:
        pb.I_ch[name] = state["I_ch"][name][fg]
    if "phi" in state:
        pb.phi = state["phi"][cg]
    if kind == "emi":
        A, b, _ = ko.assemble_emi(pb, want_B=False)
    else:
        A, b = ko.assemble_knp(pb, k), ko.knp_rhs(pb, k)
    A = A.tocsr()[:no * nd]
    gcol = (cg[:, None] * nd + np.arange(nd)[None, :]).ravel()
    A = sp.csr_matrix((A.data, gcol[A.indices], A.indptr), shape=(no * nd, state["ndof"]))
    return rank, A, np.asarray(b).ravel()[:no * nd], cg[:no]


def cpu_baseline(resolution=1):
    """CPU restatement of ONE splitting step of the same workload class, timed on this host: the oracle's assembled-CSR forms
    (reference: solver.py:270-403, 534-663) on the 4-axon mesh at `resolution` (r=1: 124 416 tets, 1.49 M DoFs), PETSc-like
    solves -- scipy CG (rtol 1e-5) and GMRES(30) (rtol 1e-7, solver.py:425-444, 684-701) -- preconditioned with the SAME
    auxiliary-space operator the GPU applies (block-Jacobi + V-cycle of the product's smoothed-aggregation hierarchy,
    oracle/cpu_precond.py), so that iteration counts are comparable.  Reports assemble time, solve time and iterations separately,
    the three quantities the reference logs per step (solver.py:499-525, 745-784).  kind = "port": FEniCS itself is not
    installable here.
    Round 4: the host's cores are used the way an MPI run of the reference would use them -- the mesh is cut into x-slabs, one
    process per slab assembles the rows of its cells (the oracle's own functions on the owned + ghost sub-mesh), the sparse products
    of the Krylov solves run on the same number of threads (knp_host_spmv); `cores` = processes / threads actually used,
    KNP_CPU_BASELINE_CORES overrides (1 = the single-process run of rounds 1-3)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = int(os.environ.get("KNP_CPU_BASELINE_CORES", min(16, avail)))
    if cores <= 1:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):                 # BLAS / LAPACK pinned to one thread: `cores` = 1 is what really ran
            return _cpu_baseline_one_core(resolution)
    # BLAS pinned to one thread: left alone, OpenBLAS starts one thread per LOGICAL core of the host (256 on the GPU boxes) for every small
    # dense product of the preconditioner and the solves spend their time in thread management (solves of the r=1 step: 4.3 s with 256
    # BLAS threads, 1.2 s with 16, 0.94 s with 1).  KNP_CPU_BASELINE_BLAS overrides; assembly processes and SpMV threads are `cores` either way.
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=int(os.environ.get("KNP_CPU_BASELINE_BLAS", 1))):
        return _cpu_baseline_parallel(resolution, cores)


def _cpu_baseline_parallel(resolution, cores):
    import multiprocessing as mp
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import knpemi_oracle as ko
    import membrane_oracle as mo
    from cpu_precond import aux_space_preconditioner
    from knpemidg import amg, _abi
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(resolution)
    pb = ko.build_idealized(m, s.array(), f.array())
    nd, ndof = pb.nd, pb.ndof
    # spawn, not fork: this process has initialised the GPU
    pool = mp.get_context("spawn").Pool(cores, initializer=_cpu_part_init, initargs=(resolution, cores))
    try:
        E = {ion["name"]: ko.nernst(pb, k) for k, ion in enumerate(pb.ions)}
        models = [mo.MembraneOracle(pb, 1, True, pb.C_M), mo.MembraneOracle(pb, 2, False, pb.C_M)]
        t0 = time.perf_counter()
        mo.oracle_membrane_step(pb, E, models, 0, pb.dt, {"stim_amplitude": 10.0}, lambda x: x[0] < 20.0e-6)
        t_ode = time.perf_counter() - t0
        t0 = time.perf_counter()
        cs = amg.ConformingSpace(m, f.array(), (1, 2))
        lv_emi = amg.build_hierarchy(cs.stiffness(pb.kappa(), membrane=(pb.mem, pb.C_phi)), psmooth=3, level0_degree=0)
        D_mean = np.mean([ion["D"] for ion in pb.ions[:-1]], axis=0)
        lv_knp = amg.build_hierarchy(cs.stiffness(D_mean, mass_coef=np.full(m.num_cells(), 1.0 / pb.dt)), psmooth=2, level0_degree=1)
        t_setup = time.perf_counter() - t0
        # mesh parts + oracle problems of the workers (setup, like the GPU's); a worker that cannot start must not hang the bench
        pool.map_async(_cpu_part_warm, range(cores)).get(timeout=600)
        state = {"phi_M": pb.phi_M, "I_ch": pb.I_ch, "ndof": ndof}

        def assemble(kind, k=0):
            parts = pool.map_async(_cpu_part_assemble, [(kind, r, k, state) for r in range(cores)]).get(timeout=900)
            rows = np.concatenate([(cg[:, None] * nd + np.arange(nd)[None, :]).ravel() for _, _, _, cg in parts])
            A = sp.vstack([A_ for _, A_, _, _ in parts], format="csr")
            b = np.concatenate([b_ for _, _, b_, _ in parts])
            perm = np.argsort(rows)                                    # part-major rows -> global row order
            return A[perm].tocsr(), b[perm]

        def threaded(A):
            A = A.tocsr()
            ip, ix, dv = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
            lib = _abi.load()

            def mv(x):
                x = np.ascontiguousarray(x, dtype=np.float64).ravel()
                y = np.empty(A.shape[0])
                lib.knp_host_spmv(A.shape[0], _abi._p(ip, _abi._i32p), _abi._p(ix, _abi._i32p), _abi._p(dv, _abi._f64p), _abi._p(x, _abi._f64p),
                                  _abi._p(y, _abi._f64p), cores)
                return y
            return spla.LinearOperator(A.shape, matvec=mv, dtype=np.float64)
        t0 = time.perf_counter()
        A, b = assemble("emi")
        t_ass_emi = time.perf_counter() - t0
        t0 = time.perf_counter()
        M = aux_space_preconditioner(A, nd, cs.dof, lv_emi)
        it_emi = [0]
        b = b - b.mean()
        x, info = spla.cg(threaded(A), b, x0=pb.phi.ravel(), rtol=1e-5, atol=0.0, maxiter=500, M=M,
                          callback=lambda xk: it_emi.__setitem__(0, it_emi[0] + 1))
        assert info == 0, "CPU baseline: EMI CG did not converge"
        pb.phi = x.reshape(-1, nd)
        t_sol_emi = time.perf_counter() - t0
        state["phi"] = pb.phi
        t_ass_knp = t_sol_knp = 0.0
        it_knp = []
        out = np.zeros_like(pb.c)
        for k in range(pb.N_ions):
            t0 = time.perf_counter()
            Ak, bk = assemble("knp", k)
            t_ass_knp += time.perf_counter() - t0
            t0 = time.perf_counter()
            Mk = aux_space_preconditioner(Ak, nd, cs.dof, lv_knp)
            it = [0]
            xk, info = spla.gmres(threaded(Ak), bk, x0=pb.c[k].ravel(), rtol=1e-7, atol=0.0, restart=30, maxiter=200, M=Mk,
                                  callback=lambda r: it.__setitem__(0, it[0] + 1), callback_type="pr_norm")
            assert info == 0, "CPU baseline: KNP GMRES did not converge"
            out[k] = xk.reshape(-1, nd)
            it_knp.append(it[0])
            t_sol_knp += time.perf_counter() - t0
        pb.c = out
        t0 = time.perf_counter()
        ko.update_phi_M(pb); ko.update_c_elim(pb)
        for k in range(len(pb.ions)):
            ko.nernst(pb, k)
        t_upd = time.perf_counter() - t0
    finally:
        pool.terminate()
        pool.join()
    step = t_ode + t_ass_emi + t_sol_emi + t_ass_knp + t_sol_knp + t_upd
    dofs = ndof * (1 + pb.N_ions)
    return {"value": dofs / step, "unit": "DoF/s", "cores": cores, "host_cores": os.cpu_count(), "kind": "port",
            "assemble_s": t_ass_emi + t_ass_knp, "solve_s": t_sol_emi + t_sol_knp, "ode_s": t_ode, "precond_setup_s": t_setup,
            "emi_iters": it_emi[0], "knp_iters": it_knp,
            "sample": "ONE splitting step on the 4-axon mesh r=%d (%d tets, %d P1-DG DoFs), %.1f s: CSR assembly by %d processes, one x-slab "
                      "each, %.1f s + scipy CG rtol 1e-5 (%d its) / GMRES(30) rtol 1e-7 (%s its) with %d-thread sparse products (BLAS pinned to one thread) %.1f s, "
                      "preconditioned with the product's auxiliary-space AMG hierarchy applied in numpy (its one-off setup, %.1f s, and "
                      "the workers' start-up are not in the step), membrane ODEs by LSODA %.1f s (one process); %d of the host's %d cores; "
                      "CPU restatement, not FEniCS"
                      % (resolution, m.num_cells(), dofs, step, cores, t_ass_emi + t_ass_knp, it_emi[0], it_knp, cores,
                         t_sol_emi + t_sol_knp, t_setup, t_ode, cores, os.cpu_count() or 0)}


def _cpu_baseline_one_core(resolution):
    import scipy.sparse.linalg as spla
    import knpemi_oracle as ko
    import membrane_oracle as mo
    from cpu_precond import aux_space_preconditioner
    from knpemidg import amg
    from knpemidg.mesh import make_mesh_3D
    m, s, f = make_mesh_3D(resolution)
    pb = ko.build_idealized(m, s.array(), f.array())
    # one membrane step with the stimulus (oracle LSODA) so that the PDE step has real work
    E = {ion["name"]: ko.nernst(pb, k) for k, ion in enumerate(pb.ions)}
    models = [mo.MembraneOracle(pb, 1, True, pb.C_M), mo.MembraneOracle(pb, 2, False, pb.C_M)]
    t0 = time.perf_counter()
    mo.oracle_membrane_step(pb, E, models, 0, pb.dt, {"stim_amplitude": 10.0}, lambda x: x[0] < 20.0e-6)
    t_ode = time.perf_counter() - t0
    # preconditioner setup (host part of the product's setup; the GPU run pays the same)
    t0 = time.perf_counter()
    cs = amg.ConformingSpace(m, f.array(), (1, 2))
    lv_emi = amg.build_hierarchy(cs.stiffness(pb.kappa(), membrane=(pb.mem, pb.C_phi)), psmooth=3, level0_degree=0)
    D_mean = np.mean([ion["D"] for ion in pb.ions[:-1]], axis=0)
    lv_knp = amg.build_hierarchy(cs.stiffness(D_mean, mass_coef=np.full(m.num_cells(), 1.0 / pb.dt)), psmooth=2, level0_degree=1)
    t_setup = time.perf_counter() - t0
    # step I: assemble + solve EMI
    t0 = time.perf_counter()
    A, b, _ = ko.assemble_emi(pb, want_B=False)
    t_ass_emi = time.perf_counter() - t0
    t0 = time.perf_counter()
    M = aux_space_preconditioner(A, pb.nd, cs.dof, lv_emi)
    it_emi = [0]
    b = b - b.mean()
    x, info = spla.cg(A, b, x0=pb.phi.ravel(), rtol=1e-5, atol=0.0, maxiter=500, M=M, callback=lambda xk: it_emi.__setitem__(0, it_emi[0] + 1))
    assert info == 0, "CPU baseline: EMI CG did not converge"
    pb.phi = x.reshape(-1, pb.nd)
    t_sol_emi = time.perf_counter() - t0
    # step II: assemble + solve KNP (two species)
    t_ass_knp = t_sol_knp = 0.0
    it_knp = []
    out = np.zeros_like(pb.c)
    for k in range(pb.N_ions):
        t0 = time.perf_counter()
        Ak = ko.assemble_knp(pb, k)
        bk = ko.knp_rhs(pb, k)
        t_ass_knp += time.perf_counter() - t0
        t0 = time.perf_counter()
        Mk = aux_space_preconditioner(Ak, pb.nd, cs.dof, lv_knp)
        it = [0]
        xk, info = spla.gmres(Ak, bk, x0=pb.c[k].ravel(), rtol=1e-7, atol=0.0, restart=30, maxiter=200, M=Mk,
                              callback=lambda r: it.__setitem__(0, it[0] + 1), callback_type="pr_norm")
        assert info == 0, "CPU baseline: KNP GMRES did not converge"
        out[k] = xk.reshape(-1, pb.nd)
        it_knp.append(it[0])
        t_sol_knp += time.perf_counter() - t0
    pb.c = out
    t0 = time.perf_counter()
    ko.update_phi_M(pb); ko.update_c_elim(pb)
    for k in range(len(pb.ions)):
        ko.nernst(pb, k)
    t_upd = time.perf_counter() - t0
    step = t_ode + t_ass_emi + t_sol_emi + t_ass_knp + t_sol_knp + t_upd
    dofs = pb.ndof * (1 + pb.N_ions)
    return {"value": dofs / step, "unit": "DoF/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
            "assemble_s": t_ass_emi + t_ass_knp, "solve_s": t_sol_emi + t_sol_knp, "ode_s": t_ode, "precond_setup_s": t_setup,
            "emi_iters": it_emi[0], "knp_iters": it_knp,
            "sample": "ONE splitting step on the 4-axon mesh r=%d (%d tets, %d P1-DG DoFs), %.1f s: assemble CSR %.1f s + "
                      "scipy CG rtol 1e-5 (%d its) / GMRES(30) rtol 1e-7 (%s its) %.1f s, preconditioned with the product's "
                      "auxiliary-space AMG hierarchy applied in numpy (its one-off setup, %.1f s, is not in the step), membrane "
                      "ODEs by LSODA %.1f s; one thread (scipy sparse kernels, BLAS pinned) on a %d-core host; CPU restatement, not FEniCS"
                      % (resolution, m.num_cells(), dofs, step, t_ass_emi + t_ass_knp, it_emi[0], it_knp, t_sol_emi + t_sol_knp,
                         t_setup, t_ode, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=2)
    ap.add_argument("--degree", type=int, default=1, help="DG degree: 1 = headline config (configs[3] mesh), 2 = configs[2] (use --resolution 1)")
    ap.add_argument("--workload", choices=["idealized", "emix"], default="idealized",
                    help="idealized = the headline 4-axon BoxMesh (default); emix = BASELINE configs[4], the reference's bundled tissue "
                         "reconstruction (121 617 unstructured tets, coordinate-path kernels, cm / ms / mV)")
    ap.add_argument("--refine", type=int, default=0, help="--workload emix: regular refinements of the tissue mesh (1: 972 936 tets)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    shm_ranks = bool(os.environ.get("KNP_COMM_SHM"))      # validation only: several ranks on ONE GPU over the shared-memory communicator
    if shm_ranks:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    # KNP_FORCE_COMM=1 takes the distributed code path (process group, slab partition, RCCL communicator) with one rank
    force_dist = os.environ.get("KNP_FORCE_COMM", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shm_ranks:
            dist.init_process_group("gloo")                # RCCL refuses two ranks on one device; the solver's own traffic goes through shm
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from idealized_common import make_solver, solver_parameters, Constant
    from knpemidg import _abi as A

    t_setup = time.perf_counter()
    if world > 1:
        # a peer exchange that never completes (the multi-GPU path has only ever run on one GPU) must not hang the node: leave with an
        # error instead of waiting for the caller's limit
        import threading
        limit = float(os.environ.get("KNP_BENCH_WATCHDOG_S", "900"))

        def _abort():
            print("[bench] rank %d: no result after %.0f s -- aborting (KNP_BENCH_WATCHDOG_S)" % (rank, limit), file=sys.stderr, flush=True)
            os._exit(3)
        wd = threading.Timer(limit, _abort)
        wd.daemon = True
        wd.start()

    def progress(msg):
        # setup of the larger meshes takes minutes of host work (mesh tables, AMG hierarchies): keep stderr alive
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.perf_counter() - t_setup, msg), file=sys.stderr, flush=True)

    r = args.resolution
    emix = args.workload == "emix"
    if emix:
        sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
        import emix_common
        if world > 1 or force_dist:
            S = emix_common.make_distributed_solver(rank, world, local_rank, dist, degree=args.degree,
                                                    mesh_tuple=emix_common.load_mesh(refine=args.refine) if args.refine else None)
        else:
            S = emix_common.make_solver(degree=args.degree, refine=args.refine)
        sp = emix_common.solver_parameters()
    elif world > 1 or force_dist:
        from knpemidg.partition import make_distributed_solver
        S = make_distributed_solver(dim=3, resolution=r, rank=rank, world=world, local_rank=local_rank, dist=dist, degree=args.degree)
    else:
        S = make_solver(dim=3, resolution=r, verbose=False, degree=args.degree)
    progress("mesh, device context and membrane models ready (%d local cells)" % S.dev.nc)
    if not emix:
        sp = solver_parameters(3, r)
    S._unpack_solver_params(sp)
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp()
    S.setup_solver_emi()
    progress("EMI preconditioner built")
    S.setup_solver_knp()
    progress("KNP preconditioners built")
    t = Constant(0.0)
    nc_global = S.global_num_cells if hasattr(S, "global_num_cells") else S.mesh.num_cells()
    dofs = nc_global * S.nd * (1 + S.N_ions)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        S.dev.sync()

    k = 0
    t_first_step = time.perf_counter() - t_setup
    for _ in range(args.warmup):
        S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
    barrier()
    progress("warm-up done")
    t_ready = time.perf_counter() - t_setup
    timers0 = (S.emi_solve_timer, S.knp_solve_timer, S.emi_ass_timer + S.knp_ass_timer, S.ode_solve_timer)   # config.*_s cover the timed steps only
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
        if os.environ.get("KNP_BENCH_STEP_TIMES"):
            progress("step %d enqueued/solved in %.2f ms" % (k, 1e3 * (time.perf_counter() - ts)))
    barrier()
    elapsed = time.perf_counter() - t0
    timers1 = (S.emi_solve_timer, S.knp_solve_timer, S.emi_ass_timer + S.knp_ass_timer, S.ode_solve_timer)
    emi_s, knp_s, ass_s, ode_s = (b - a for a, b in zip(timers0, timers1))
    its_emi = list(S.emi_niter[-args.steps:])
    its_knp = [max(n) for n in S.knp_niter[-args.steps:]]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shm_ranks else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = 1e3 * elapsed / args.steps

    # --- roofline of the dominant kernels (operator applies), HIP events on the context's stream.
    # (1) in-solver: every apply launched by the solves of two further steps is bracketed by an event pair (knp_apply_timing);
    # (2) back-to-back: 200 launches rotating three vector pairs (knp_bench_apply) -- the figure `roofline.achieved` uses is
    #     the SLOWER, in-solver one.
    nc_local = S.dev.nc_owned
    S.dev.apply_timing(True)
    for _ in range(2):
        S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
    emi_solver_ms, emi_n = S.dev.apply_timing_read(0)
    knp_solver_ms, knp_n = S.dev.apply_timing_read(1)
    S.dev.apply_timing(False)
    rng = np.random.default_rng(0)
    S.dev.upload(A.F_X, rng.uniform(-1, 1, size=S.dev.size(A.F_X)))
    S.dev.update_kappa(); S.dev.update_dnphi()
    emi_b2b_ms = S.dev.bench_apply(0, 200)
    knp_b2b_ms = S.dev.bench_apply(1, 200)
    emi_ms = max(emi_solver_ms, emi_b2b_ms) if emi_n else emi_b2b_ms
    knp_ms = max(knp_solver_ms, knp_b2b_ms) if knp_n else knp_b2b_ms
    # algorithmic bytes per cell: P1 137 / 217, P2 281 / 457
    emi_bpc = EMI_BYTES_PER_CELL if args.degree == 1 else 281.0
    knp_bpc = KNP_BYTES_PER_CELL if args.degree == 1 else 457.0
    emi_gbs = emi_bpc * nc_local / (emi_ms * 1e-3) / 1e9
    knp_gbs = knp_bpc * nc_local / (knp_ms * 1e-3) / 1e9

    # HBM-side traffic of the same kernel on the same workload, from the committed rocprofv3 --pmc passes
    # (counters cannot be read from inside this process); null when the workload differs
    cls = bool(S.dev.n_geometry_classes)
    if args.degree == 1:
        emi_name = {3: "k_emi_apply_ring", 1: "k_emi_apply_cls_staged<3,256>", 10: "k_emi_apply_ring_u"}.get(S.dev.apply_variant(0), "k_emi_apply<3,3>")
        kv = S.dev.apply_variant(1)
        knp_name = {7: "k_knp_apply_ring<2>", 2: "k_knp_apply_halo<2,false>", 6: "k_knp_apply_halo<2,true>",
                    1: "k_knp_apply_cls_staged<3,2,256>", 10: "k_knp_apply_ring_u<2>"}.get(kv, "k_knp_apply<3,2>")
    else:
        emi_name = "k_emi_apply_p2<3,256,%s>" % ("true" if cls else "false")
        knp_name = "k_knp_apply_p2<3,256,%s>" % ("true" if cls else "false")
    # FP64 vector instructions per cell of the two apply kernels (static counts from the gfx950 ISA, loops fully unrolled: DESIGN.md
    # sections 4.0 / 4b) against the chip's FP64 issue peak, 256 CUs x 4 SIMDs x 16 lanes per cycle at 2.4 GHz: the second roofline
    # of these kernels (the P2 applies are bound by it, the P1 ring-staged applies sit between it and the HBM one)
    # (tools/count_fp64.py prints them from the current sources; the DG-P2 KNP kernel runs one species per pass: 2 x 2304)
    fp64_per_cell = {"k_emi_apply_ring": 542, "k_knp_apply_ring<2>": 435, "k_emi_apply_p2<3,256,true>": 2035,
                     "k_knp_apply_p2<3,256,true>": 2 * 2304, "k_emi_apply_ring_u": 636, "k_knp_apply_ring_u<2>": 684,
                     "k_emi_apply_p2<3,256,false>": 2264, "k_knp_apply_p2<3,256,false>": 2 * 2533}
    FP64_PEAK_TINST = 256 * 4 * 16 * 2.4e9 / 1e12

    def fp64_issue(name, ms):
        n = fp64_per_cell.get(name)
        if n is None:
            return None
        ach = n * nc_local / (ms * 1e-3) / 1e12
        return {"fp64_instructions_per_cell": n, "achieved": ach, "peak": FP64_PEAK_TINST, "unit": "T lane-instructions/s",
                "frac": ach / FP64_PEAK_TINST}
    # A record counts only for the code it was measured on: it carries the sha256 (first 16 hex digits) of the kernel's source file at the
    # time of the counter pass (tools/pmc_traffic_update.py), and a kernel edited since then reports null until a new pass is committed.
    traffic = traffic_emi = None
    try:
        import hashlib
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        sha = {}

        def current(src):
            if src not in sha:
                with open(os.path.join(ROOT, "knp-emi-dg_amd", "csrc", src), "rb") as fh:
                    sha[src] = hashlib.sha256(fh.read()).hexdigest()[:16]
            return sha[src]
        for key, rec in pmc.items():
            if isinstance(rec, dict) and rec.get("cells_per_launch") == nc_local and rec.get("source") and rec.get("source_sha16") == current(rec["source"]):
                if key.split(":")[0] == emi_name:
                    traffic_emi = rec["traffic_bytes"]
                if key.split(":")[0] == knp_name:
                    traffic = rec["traffic_bytes"]
    except (OSError, ValueError, KeyError):
        pass

    progress("timed region: %.2f ms/step; a %d-step run of examples/idealized-geometries/run_3D.py:60-62 at this rate + the %.1f s of setup "
             "above = %.1f s end to end" % (ms_per_step, 200, t_ready, t_ready + 0.2 * ms_per_step))
    if rank == 0:
        out = {
            "metric": "DoF-updates/sec per PDE timestep", "value": dofs * args.steps / elapsed, "unit": "DoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("EMIx tissue reconstruction volume_ncells_5_size_5000%s (%d unstructured tets, %d P%d-DG DoFs), glial + "
                                    "neuronal membranes + stimulus, full splitting step" % (
                                        ", %d regular refinement(s)" % args.refine if args.refine else "", nc_global, dofs, args.degree)) if emix else
                                   ("3D idealized 4-axon mesh r=%d (%d tets, %d P%d-DG DoFs: phi + K,Cl solved, Na eliminated), "
                                    "HH membranes + stimulus, full splitting step" % (r, nc_global, dofs, args.degree)),
                       "parallelism": ("rcb%d" if emix else "slab%d") % world,
                       "cpu_baseline_workload": "the same mesh family at r=1 (124 416 tets, 1 492 992 DoFs), ONE step on cpu_baseline.cores host "
                                                "cores -- NOT the GPU line's r=%d mesh: one oracle step at r=2 is ~70 s of assembly + solves "
                                                "on one core, beyond the bounded sample the default run may spend" % r,
                       "preconditioner": ("cell-block-Jacobi + conforming-P%d auxiliary space, smoothed-aggregation AMG V-cycle; block inverses, level "
                                          "matrices and coarse inverse stored in fp32, vectors / operator applies / stopping tests in fp64"
                                          % args.degree) if S.use_amg
                       else "cell-block-Jacobi",
                       "emi_iters_per_step": float(np.mean(its_emi)), "knp_iters_per_step": float(np.mean(its_knp)),
                       # host wall time of the phases over the TIMED steps only (each phase ends in a device sync)
                       "emi_solve_s": emi_s, "knp_solve_s": knp_s, "assemble_s": ass_s, "ode_s": ode_s,
                       "setup_s_before_first_step": t_first_step,
                       # DG-level smoother of the EMI preconditioner as measured by the run itself on its solves 1-4 (knpemidg/solver.py)
                       "emi_dg_smoother": getattr(S, "emi_dg_chebyshev_measured", None)},
            # dominant kernel of a step = the KNP operator apply (all solved species in one launch; ~30 % of the kernel time
            # of the profiled run, profiles/): the EMI apply, same design, is reported next to it
            "roofline": {"bound": "hbm", "kernel": knp_name, "achieved": knp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": knp_gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_kernel_us": knp_ms * 1e3,
                         "algorithmic_bytes_per_cell": knp_bpc, "cells_per_launch": nc_local,
                         "in_solver_us": knp_solver_ms * 1e3, "in_solver_launches": knp_n, "back_to_back_us": knp_b2b_ms * 1e3,
                         "fp64_issue": fp64_issue(knp_name, knp_ms),
                         "emi_apply": {"kernel": emi_name, "achieved": emi_gbs, "frac": emi_gbs / HBM_PEAK_GBS, "traffic": traffic_emi,
                                       "avg_kernel_us": emi_ms * 1e3, "algorithmic_bytes_per_cell": emi_bpc,
                                       "in_solver_us": emi_solver_ms * 1e3, "in_solver_launches": emi_n,
                                       "fp64_issue": fp64_issue(emi_name, emi_ms),
                                       "back_to_back_us": emi_b2b_ms * 1e3}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
