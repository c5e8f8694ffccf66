"""The partitioned solver END TO END with several ranks on one GPU: every rank is a process with its own device context on cuda:0 and
the host-staged shared-memory communicator (knp_comm_init_shm) in place of RCCL, which refuses two ranks on a device.  Everything
around the transport runs as on a multi-GPU node -- partition, ghost layer, halo tables, pack / unpack, all-reduced Krylov scalars
and restricted residuals, replicated hierarchies, membrane facets and ODE nodes on a cut -- and the result must be the single-rank
solution."""
import os
import subprocess
import sys
import uuid

import numpy as np
import pytest

from common import relerr

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _single_rank(n_axons, steps, method, degree=1):
    from common_examples import make_solver, solver_parameters, Constant
    if method == "emix":
        ex = os.path.join(os.path.dirname(HERE), "examples", "emix_simulations")
        if ex not in sys.path:
            sys.path.insert(0, ex)
        import emix_common
        S = emix_common.make_solver()
        S._unpack_solver_params(emix_common.solver_parameters()._replace(rtol_emi=1e-10, rtol_knp=1e-12))
    else:
        S = make_solver(dim=3, resolution=0, n_axons=n_axons, degree=degree)
        S._unpack_solver_params(solver_parameters(3, 0)._replace(rtol_emi=1e-10, rtol_knp=1e-12))
    S.save_fields = S.save_solver_stats = False
    S.splitting_scheme = True
    S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
    t = Constant(0.0)
    for k in range(steps):
        S.step_membrane_models(k)
        S.solve_for_time_step(k, t)
    nc = S.mesh.num_cells()
    x = S.mesh.coords[S.mesh.cells]
    vol = np.abs(np.linalg.det(x[:, 1:] - x[:, :1])) / 6.0
    out = (S.c.array().reshape(S.N_ions, nc, S.nd).copy(), S.phi.array().reshape(nc, S.nd).copy(), vol, list(S.emi_niter),
           [max(n) for n in S.knp_niter], np.asarray(S.emi_targets))
    S.dev.close()
    return out


@pytest.mark.parametrize("world,method,n_axons,dist0", [(2, "slab", 4, 1), (3, "slab", 4, 1), (3, "rcb", 1, 1), (3, "emix", 0, 1),
                                                        (3, "thin", 4, 1), (2, "p2", 4, 1), (2, "slab", 4, 0)])
def test_partitioned_solver_with_several_ranks_on_one_gpu(hip_lib, tmp_path, monkeypatch, world, method, n_axons, dist0):
    # dist0 = 1: the finest conforming level ROW-DISTRIBUTED (sub-assembled matrices, point-to-point exchange of the shared dofs, one
    # all-reduce of the level-1 residual; csrc/amg.hip dist0) -- the default of a partitioned run; 0: replicated behind an all-reduce of
    # the level-0 residual (rounds 1-2).  The small meshes get a coarse-size limit that leaves at least one level below the finest.
    steps = 3
    monkeypatch.setenv("KNP_AMG_DIST0", str(dist0))
    if method != "emix":
        monkeypatch.setenv("KNP_AMG_MAXCOARSE", "300")
    name = "/knp_%s" % uuid.uuid4().hex[:16]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "multirank_worker.py"), str(r), str(world), name, str(tmp_path), method,
                               str(n_axons), str(steps)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            logs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-2000:] for l in logs)
    c_ref, phi_ref, vol, emi_ref, knp_ref, targets_ref = _single_rank(n_axons, steps, method, degree=2 if method == "p2" else 1)
    nc = c_ref.shape[1]
    c = np.full_like(c_ref, np.nan)
    phi = np.full_like(phi_ref, np.nan)
    seen = np.zeros(nc, dtype=int)
    targets = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))["emi_targets"] for r in range(world)]
    # the residual target of the EMI stopping test must be the SAME number on every rank (ranks that disagree about
    # convergence leave the loop of collectives at different iterations) -- bit for bit, step 0 (all-reduced load norm of the initial
    # state) included -- and the single-rank one up to the summation order
    assert len(targets[0]) == steps and all(np.array_equal(t, targets[0]) for t in targets[1:]), targets
    assert np.abs(targets[0] / targets_ref - 1.0).max() < 1e-6, (targets[0], targets_ref)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        c[:, d["cells"]] = d["c"]
        phi[d["cells"]] = d["phi"]
        seen[d["cells"]] += 1
        assert len(d["emi_its"]) == steps and d["emi_its"].max() < 1000 and d["knp_its"].max() < 1000
        assert int(d["dist0"]) == (2 if dist0 else 0), d["dist0"]              # EMI + the shared KNP hierarchy, both row-distributed
        # the partitioned preconditioner is the single-rank one up to rounding: same iteration counts
        assert np.abs(d["emi_its"] - np.asarray(emi_ref)).max() <= 1 and np.abs(d["knp_its"] - np.asarray(knp_ref)).max() <= 1, (
            d["emi_its"], emi_ref, d["knp_its"], knp_ref)
    assert (seen == 1).all()                                   # every cell owned by exactly one rank
    mean = lambda p: p - (p.mean(axis=1) * vol).sum() / vol.sum()
    assert relerr(c, c_ref) < 1e-8
    assert relerr(mean(phi), mean(phi_ref)) < 1e-6


def test_bench_contract_with_two_ranks(hip_lib):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per process), here with both ranks
    on the one GPU over the shm communicator: exactly one JSON line from rank 0, whole-job value, the fields of the contract."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, KNP_COMM_SHM="/knp_%s" % uuid.uuid4().hex[:16])
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--resolution", "0",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["config"]["parallelism"] == "slab2" and d["vs_baseline"] is None
