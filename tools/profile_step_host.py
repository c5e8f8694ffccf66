"""Host-side cProfile of the time-stepping loop (after warm-up): where the Python side of a splitting step spends its time.
usage: python tools/profile_step_host.py [emix|R] [steps]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "examples", "emix_simulations"), os.path.join(ROOT, "examples", "idealized_geometries")]
which = sys.argv[1] if len(sys.argv) > 1 else "emix"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "emix":
    import emix_common as E
    S, sp, Constant = E.make_solver(), E.solver_parameters(), E.Constant
else:
    import idealized_common as I
    S, sp, Constant = I.make_solver(dim=3, resolution=int(which)), I.solver_parameters(3, int(which)), I.Constant
S._unpack_solver_params(sp)
S.save_fields = S.save_solver_stats = False
S.splitting_scheme = True
S.setup_varform_emi(); S.setup_varform_knp(); S.setup_solver_emi(); S.setup_solver_knp()
t = Constant(0.0)
k = 0
for _ in range(5):
    S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
S.dev.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(steps):
    S.step_membrane_models(k); S.solve_for_time_step(k, t); k += 1
S.dev.sync()
pr.disable()
print("%.3f ms per step" % (1e3 * (time.perf_counter() - t0) / steps))
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
