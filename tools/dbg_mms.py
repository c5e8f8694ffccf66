import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "mms")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "knp-emi-dg_amd"))
import numpy as np
import run_MMS_space as R
from knpemidg import _abi as A
# monkeypatch Device.emi_solve to report
orig = A.Device.emi_solve
def dbg(self, rtol, atol=1e-40, maxit=1000, check_every=25):
    import ctypes as C
    it = C.c_int(0); res = np.zeros(3)
    rc = self.lib.knp_emi_solve(self.ctx, rtol, atol, maxit, check_every, C.byref(it), A._p(res, A._f64p))
    print("emi_solve rc", rc, "its", it.value, "res0 %.3e res %.3e bnorm %.3e rtol %.1e" % (res[0], res[1], res[2], rtol), flush=True)
    if rc: raise A.KnpError("fail")
    return it.value, res
A.Device.emi_solve = dbg
for noamg in ("0", "1"):
    os.environ["KNP_NO_AMG"] = noamg
    try:
        print("noamg", noamg, R.run(3, verbose=False))
    except Exception as e:
        print("failed", e)
