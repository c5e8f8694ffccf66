"""cProfile of the host-side setup on the bench workload: mesh + device context, then the preconditioner setup.
usage: profile_setup.py [resolution] [degree]"""
import cProfile, os, pstats, sys, time
t00 = time.time()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "idealized_geometries")]
os.environ["KNP_AMG_SERIAL_SETUP"] = "1"
from idealized_common import make_solver, solver_parameters
print("imports %.2f s" % (time.time() - t00))
r = int(sys.argv[1]) if len(sys.argv) > 1 else 2
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t = time.time()
pr = cProfile.Profile()
pr.enable()
S = make_solver(dim=3, resolution=r, degree=deg)
pr.disable()
print("mesh + device %.2f s" % (time.time() - t))
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
S._unpack_solver_params(solver_parameters(3, r))
S.save_fields = S.save_solver_stats = False
S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp()
pr = cProfile.Profile()
t = time.time()
pr.enable()
S.setup_solver_emi(); S.setup_solver_knp()
pr.disable()
print("preconditioner setup %.2f s" % (time.time() - t))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
