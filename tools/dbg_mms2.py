import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R_, "examples", "mms")); sys.path.insert(0, os.path.join(R_, "knp-emi-dg_amd"))
import numpy as np
import run_MMS_space as R
from knpemidg import _abi as A
import ctypes as C
def dbg(self, rtol, atol=1e-40, maxit=1000, check_every=25):
    it = C.c_int(0); res = np.zeros(3)
    rc = self.lib.knp_emi_solve(self.ctx, rtol, atol, maxit, check_every, C.byref(it), A._p(res, A._f64p))
    print("emi_solve rc", rc, "its", it.value, "res0 %.3e res %.3e bnorm %.3e rtol %.1e" % (res[0], res[1], res[2], rtol), flush=True)
    if rc: raise A.KnpError("fail")
    return it.value, res
def dbgk(self, rtol, atol=1e-40, maxit=1000, min_it=5, check_every=10):
    it = (C.c_int * self.n_sys)(); res = np.zeros(3 * self.n_sys)
    rc = self.lib.knp_knp_solve(self.ctx, rtol, atol, maxit, min_it, check_every, it, A._p(res, A._f64p))
    print("knp_solve rc", rc, list(it), res, flush=True)
    if rc: raise A.KnpError("fail knp " + self.lib.knp_last_error(self.ctx).decode())
    return list(it), res.reshape(self.n_sys, 3)
A.Device.emi_solve = dbg
A.Device.knp_solve = dbgk
for r in (3, 4, 5):
    try:
        print("r", r, R.run(r))
    except Exception as e:
        print("failed", e)
