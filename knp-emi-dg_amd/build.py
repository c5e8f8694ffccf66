"""Build libknpemi_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "knpemidg", "libknpemi_hip.so")
HOST_SOURCES = ["host_sparse.cpp"]
SOURCES = ["abi.hip", "apply_p1.hip", "rhs_p1.hip", "krylov.hip", "comm.hip", "amg.hip", "ode.hip", "tab_dg.hip", "apply_p2.hip", "apply_ring.hip", "apply_ring_u.hip"]


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(HERE, "..", "include", "knpemi_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",   # hipFree / event calls in teardown paths are fire-and-forget

               "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    # host-only setup kernels (threaded sparse products of the AMG setup): plain g++
    for s in HOST_SOURCES:
        o = os.path.join(CSRC, s.replace(".cpp", ".o"))
        cmd = [os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-pthread", "-c", os.path.join(CSRC, s), "-o", o]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(o)
    ok = True
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0 or verbose:
            sys.stderr.write("[%s]\n%s\n" % (s, out))
        ok &= p.returncode == 0
    if not ok:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-lpthread", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
