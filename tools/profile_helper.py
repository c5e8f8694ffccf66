"""Where the hierarchy helpers (knpemidg/setup_worker.py) spend their time: cProfile of the mesh build and of setup_worker.run() for the EMI
and the KNP job of the 4-axon mesh at refinement r, in this process (no GPU).  The first-step latency of a run is
mesh build -> EMI helper -> upload (bench.py stderr stamps), so these two profiles are its critical path.
usage: python tools/profile_helper.py [r] [degree]"""
import cProfile, os, pickle, pstats, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd"))
import numpy as np                                                   # noqa: E402
from knpemidg import mesh as M, setup_worker                         # noqa: E402

r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def profiled(label, fn, top=16):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = fn()
    pr.disable()
    print("---- %s: %.3f s" % (label, time.perf_counter() - t0))
    pstats.Stats(pr).sort_stats("tottime").print_stats(top)
    return out


mesh, sub, surf = profiled("mesh build", lambda: M.make_mesh_3D(r), 12)
nc = mesh.num_cells()
nd = 4 if degree == 1 else 10
kappa = np.random.default_rng(0).uniform(0.5, 1.5, (nc, nd))
job = setup_worker.emi_job(mesh, surf.array(), [1], degree, kappa, 1.0)
t0 = time.perf_counter()
blob = pickle.dumps(job, protocol=pickle.HIGHEST_PROTOCOL)
print("pickle of the EMI job: %.3f s, %.0f MB" % (time.perf_counter() - t0, len(blob) / 1e6))
res = profiled("EMI helper run", lambda: setup_worker.run(job), 26)
t0 = time.perf_counter()
blob = pickle.dumps(res, protocol=pickle.HIGHEST_PROTOCOL)
print("pickle of the EMI result: %.3f s, %.0f MB" % (time.perf_counter() - t0, len(blob) / 1e6))
D = [{0: 1.33e-9, 1: 1.33e-9}, {0: 1.96e-9, 1: 1.96e-9}, {0: 2.03e-9, 1: 2.03e-9}]
kjob = setup_worker.job_from_solver(mesh, sub.array(), surf.array(), [1], degree, D[:2], 1.0e-4, 2)
res = profiled("KNP helper run", lambda: setup_worker.run(kjob), 20)
