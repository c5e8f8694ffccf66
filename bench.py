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


def cpu_baseline(seconds_hint=20.0):
    """The oracle (CPU restatement, assembled CSR + scipy CG / GMRES with block-Jacobi) timed on this
    host for ONE splitting step of the r=0 mesh of the same workload.  kind = "port": FEniCS itself is
    not installable here."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import knpemi_oracle as ko
    from knpemidg.mesh import make_mesh_3D
    from knpemidg.models import mm_hh
    from knpemidg.membrane import integrate_batch
    m, s, f = make_mesh_3D(0)
    pb = ko.build_idealized(m, s.array(), f.array())
    # one ODE step with the stimulus so that the PDE step has real work
    mem = pb.mem
    n = len(mem)
    st = np.array([mm_hh.init_state_values() for _ in range(n)])
    pr = np.array([mm_hh.init_parameter_values() for _ in range(n)])
    P = ko.idealized_params()
    pr[:, mm_hh.parameter_indices("Cm")] = P["C_M"]
    for k, ion in enumerate(pb.ions):
        pr[:, mm_hh.parameter_indices("E_" + ion["name"])] = ko.nernst(pb, k)
    pr[:, mm_hh.parameter_indices("K_e")] = P["init"]["K"][1]
    pr[:, mm_hh.parameter_indices("Na_i")] = P["init"]["Na"][0]
    fm = m.facet_midpoints()[mem]
    pr[fm[:, 0] < 20e-6, mm_hh.parameter_indices("stim_amplitude")] = 10.0
    st, _ = integrate_batch(mm_hh.rhs, 0.0, P["dt"], st, pr)
    pb.phi_M[mem] = st[:, 3]
    for ion in pb.ions:
        pb.I_ch[ion["name"]][mem] = pr[:, mm_hh.parameter_indices("I_ch_" + ion["name"])]
    stats = {}
    t0 = time.perf_counter()
    ko.solve_for_time_step(pb, direct=False, rtol_emi=1e-5, rtol_knp=1e-7, stats=stats)
    dt = time.perf_counter() - t0
    dofs = pb.ndof * (1 + pb.N_ions)
    return {"value": dofs / dt, "unit": "DoF/s", "cores": 1, "kind": "port",
            "sample": "one splitting step (assemble CSR + scipy CG rtol 1e-5 + GMRES(30) rtol 1e-7, block-Jacobi) "
                      "on the r=0 mesh of the same geometry: 15552 tets, %d DoFs, %.1f s, EMI its %s, KNP its %s; "
                      "CPU restatement, not FEniCS" % (dofs, dt, stats.get("emi_iters"), stats.get("knp_iters"))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--resolution", type=int, default=2)
    ap.add_argument("--degree", type=int, default=1, help="DG degree: 1 = headline config (configs[3] mesh), 2 = configs[2] (use --resolution 1)")
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
    torch.cuda.set_device(local_rank)
    dist = None
    # KNP_FORCE_COMM=1 takes the distributed code path (process group, slab partition, RCCL communicator) with one rank
    force_dist = os.environ.get("KNP_FORCE_COMM", "0") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from idealized_common import make_solver, solver_parameters, Constant
    from knpemidg import _abi as A

    t_setup = time.perf_counter()

    def progress(msg):
        # setup of the larger meshes takes minutes of host work (mesh tables, AMG hierarchies): keep stderr alive
        if rank == 0:
            print("[bench %6.1fs] %s" % (time.perf_counter() - t_setup, msg), file=sys.stderr, flush=True)

    r = args.resolution
    if world > 1 or force_dist:
        from knpemidg.partition import make_distributed_solver
        S = make_distributed_solver(dim=3, resolution=r, rank=rank, world=world, local_rank=local_rank, dist=dist, degree=args.degree)
    else:
        S = make_solver(dim=3, resolution=r, verbose=False, degree=args.degree)
    progress("mesh, device context and membrane models ready (%d local cells)" % S.dev.nc)
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
    for _ in range(args.warmup):
        S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
    barrier()
    progress("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
        if os.environ.get("KNP_BENCH_STEP_TIMES"):
            progress("step %d enqueued/solved in %.2f ms" % (k, 1e3 * (time.perf_counter() - ts)))
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
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
        emi_name = "k_emi_apply_cls_staged<3,256>" if cls else "k_emi_apply<3,3>"
        knp_name = "k_knp_apply_cls_staged<3,2,256>" if cls else "k_knp_apply<3,2>"
    else:
        emi_name = "k_emi_apply_p2<3,256,%s>" % ("true" if cls else "false")
        knp_name = "k_knp_apply_p2<3,256,%s>" % ("true" if cls else "false")
    traffic = traffic_emi = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pmc.get(emi_name, {}).get("cells_per_launch") == nc_local:
            traffic_emi = pmc[emi_name]["traffic_bytes"]
        if pmc.get(knp_name, {}).get("cells_per_launch") == nc_local:
            traffic = pmc[knp_name]["traffic_bytes"]
    except (OSError, ValueError):
        pass

    if rank == 0:
        out = {
            "metric": "DoF-updates/sec per PDE timestep", "value": dofs * args.steps / elapsed, "unit": "DoF/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D idealized 4-axon mesh r=%d (%d tets, %d P%d-DG DoFs: phi + K,Cl solved, Na eliminated), "
                                   "HH membranes + stimulus, full splitting step" % (r, nc_global, dofs, args.degree),
                       "parallelism": "slab%d" % world,
                       "preconditioner": ("cell-block-Jacobi + conforming-P%d auxiliary space, smoothed-aggregation AMG V-cycle" % args.degree) if S.use_amg
                       else "cell-block-Jacobi",
                       "emi_iters_per_step": float(np.mean(S.emi_niter[-args.steps:])),
                       "knp_iters_per_step": float(np.mean([max(n) for n in S.knp_niter[-args.steps:]])),
                       "emi_solve_s": S.emi_solve_timer, "knp_solve_s": S.knp_solve_timer,
                       "assemble_s": S.emi_ass_timer + S.knp_ass_timer, "ode_s": S.ode_solve_timer},
            # dominant kernel of a step = the KNP operator apply (all solved species in one launch; ~30 % of the kernel time
            # of the profiled run, profiles/): the EMI apply, same design, is reported next to it
            "roofline": {"bound": "hbm", "kernel": knp_name, "achieved": knp_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": knp_gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_kernel_us": knp_ms * 1e3,
                         "algorithmic_bytes_per_cell": knp_bpc, "cells_per_launch": nc_local,
                         "in_solver_us": knp_solver_ms * 1e3, "in_solver_launches": knp_n, "back_to_back_us": knp_b2b_ms * 1e3,
                         "emi_apply": {"kernel": emi_name, "achieved": emi_gbs, "frac": emi_gbs / HBM_PEAK_GBS, "traffic": traffic_emi,
                                       "avg_kernel_us": emi_ms * 1e3, "algorithmic_bytes_per_cell": emi_bpc,
                                       "in_solver_us": emi_solver_ms * 1e3, "in_solver_launches": emi_n,
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
