"""Helper processes of the preconditioner setup: one builds the KNP hierarchies (reference: BoomerAMG on AA_knp, solver.py:688, 767),
one the first EMI hierarchy (BoomerAMG on BB_emi, solver.py:433, 505) from the initial state, while the parent process creates the
device context.

Why a process: two host threads building hierarchies side by side are SLOWER than one after the other (numpy's short GIL-holding
calls convoy and the threaded LAPACK / sparse-product pools oversubscribe the cores: 0.90 s serial vs 1.44 s threaded on the r=1
mesh).  The helper never touches the GPU (the device list is hidden from it) and is started as an ordinary child process.

protocol: one pickled job on stdin -> one pickled {"groups": [(members, levels)]} or {"error": text} on stdout."""
import os
import pickle
import sys
import types


# ---- transport of jobs and results ------------------------------------------------------------------------------------------------
# The job of the r=3 mesh is 830 MB of arrays and its result 690 MB: pickled through the pipes that was 0.9 + 0.7 s (two copies in the
# pickle, two through the 64 KB pipe buffer).  The arrays now travel OUT OF BAND (pickle protocol 5): the sender writes them once into a
# file under /dev/shm, the pipe carries the small pickle with the file's name, and the receiver maps the file copy-on-write and
# unlinks it -- its arrays are views of the mapping.  No /dev/shm (or any failure on the way): everything in band, as before.
_SHM_DIR = "/dev/shm"
_SHM_MIN = 1 << 20            # payloads below 1 MB stay in band
_SHM_ALIGN = 64
_shm_made = []                # files this process wrote and whose receiver may not have unlinked yet (removed at exit)
_shm_count = [0]


def _dump(obj, out, name=None):
    """name: the file to use (the helper writes its result into a file its PARENT named and will remove whatever happens to the helper)."""
    bufs = []
    try:
        payload = pickle.dumps(obj, protocol=5, buffer_callback=bufs.append)
        raws = [b.raw() for b in bufs]
        total = sum(r.nbytes for r in raws)
        if total < _SHM_MIN or not os.path.isdir(_SHM_DIR) or os.environ.get("KNP_SETUP_NO_SHM", "0") == "1":
            raise OSError("in band")
        if not _shm_made:
            import atexit
            atexit.register(_shm_cleanup)
        if name is None:
            name = _new_name()
        with open(name, "wb") as f:
            _shm_made.append(name)
            for r in raws:
                f.write(r)
                f.write(b"\0" * (-r.nbytes % _SHM_ALIGN))                 # every array starts on a 64-byte boundary of the mapping
        head = {"payload": payload, "file": name, "sizes": [r.nbytes for r in raws]}
    except (OSError, ValueError, BufferError):
        if name:                                      # a file that could not be written completely (/dev/shm too small): remove it now
            try:
                os.unlink(name)
            except OSError:
                pass
            try:
                _shm_made.remove(name)
            except ValueError:
                pass
        head = {"inband": pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)}
    pickle.dump(head, out, protocol=pickle.HIGHEST_PROTOCOL)
    out.flush()
    return head.get("file")


def _new_name():
    _shm_count[0] += 1
    return os.path.join(_SHM_DIR, "knp_setup_%d_%d" % (os.getpid(), _shm_count[0]))


def _load(inp):
    head = pickle.load(inp)
    if "inband" in head:
        return pickle.loads(head["inband"])
    import mmap
    name, sizes = head["file"], head["sizes"]
    try:
        with open(name, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY) if sum(sizes) else None     # private, writable: in-place edits stay local
    finally:
        try:
            os.unlink(name)
        except OSError:
            pass
    view, off, bufs = memoryview(mm) if mm is not None else memoryview(b""), 0, []
    for n in sizes:
        bufs.append(view[off:off + n])
        off += n + (-n % _SHM_ALIGN)
    return pickle.loads(head["payload"], buffers=bufs)


def _remove(name):
    if name:
        try:
            os.unlink(name)
        except OSError:
            pass
        try:
            _shm_made.remove(name)
        except ValueError:
            pass


def _shm_cleanup():
    while _shm_made:
        try:
            os.unlink(_shm_made.pop())
        except OSError:
            pass


def mesh_stub(coords, cells, facet_cells, facets=None, facet_local=None):
    """The attributes of a Mesh that the conforming spaces read (facets / facet_local: only the membrane term of the EMI operator)."""
    m = types.SimpleNamespace(coords=coords, cells=cells, facet_cells=facet_cells, facets=facets, facet_local=facet_local,
                              gdim=coords.shape[1])
    m.num_cells = lambda: cells.shape[0]
    return m


def emi_job(mesh, facet_tags, membrane_tags, degree, kappa, C_phi):
    """kappa: [nc, nd] nodal values the hierarchy is built from (the initial state)."""
    return {"kind": "emi", "coords": mesh.coords, "cells": mesh.cells, "facet_cells": mesh.facet_cells, "facets": mesh.facets,
            "facet_local": mesh.facet_local, "facet_tags": facet_tags, "membrane_tags": list(membrane_tags), "degree": int(degree),
            "kappa": kappa, "C_phi": float(C_phi)}


def job_from_solver(gmesh, sub_tags, facet_tags, membrane_tags, degree, D_subs, dt, level0_degree):
    return {"coords": gmesh.coords, "cells": gmesh.cells, "facet_cells": gmesh.facet_cells, "sub_tags": sub_tags, "facet_tags": facet_tags,
            "membrane_tags": list(membrane_tags), "degree": int(degree), "D_subs": [dict((int(k), float(v)) for k, v in d.items()) for d in D_subs],
            "dt": float(dt), "level0_degree": int(level0_degree)}


def run(job):
    from knpemidg import amg
    mesh = mesh_stub(job["coords"], job["cells"], job["facet_cells"], job.get("facets"), job.get("facet_local"))
    cs = amg.ConformingSpace(mesh, job["facet_tags"], job["membrane_tags"])
    cs2 = amg.ConformingSpaceP2(cs) if job["degree"] != 1 else None
    if job.get("kind") == "emi":
        levels = amg.build_emi_levels(cs, cs2, job["facet_tags"], job["membrane_tags"], job["kappa"], job["C_phi"])
        return {"levels": levels, "dof": (cs2 if cs2 is not None else cs).dof}
    return amg.build_knp_groups(cs, cs2, job["sub_tags"], job["D_subs"], job["dt"], job["level0_degree"])


def main():
    out = sys.stdout.buffer
    sys.stdout = sys.stderr                       # nothing but the result may reach the pipe
    job = None
    try:
        # everything the job needs is imported BEFORE the job arrives: a helper started ahead of time (prestart) spends the
        # interpreter / numpy / scipy / library start-up (0.6 s) while the parent is still building its mesh
        import numpy, scipy.sparse, scipy.sparse.csgraph, scipy.linalg                    # noqa: F401, E401
        from knpemidg import amg, _abi                                                     # noqa: F401
        try:
            _abi.load()
        except Exception:
            pass
        _abi._stamp("helper: modules imported, waiting for the job")
        job = _load(sys.stdin.buffer)
        _abi._stamp("helper: job received (%s)" % job.get("kind", "knp"))
        res = {"groups": run(job)}
        _abi._stamp("helper: hierarchy built (%s)" % job.get("kind", "knp"))
    except BaseException as e:                    # reported to the parent, which falls back to building in-process
        import traceback
        res = {"error": "%s\n%s" % (e, traceback.format_exc())}
    _dump(res, out, name=(job or {}).get("_result_file") if isinstance(job, dict) else None)
    try:
        _abi._stamp("helper: result sent")
    except Exception:
        pass
    # the parent unlinks the result file once it has mapped it; should it die first, the file goes when this process is reaped
    # (kill -> no atexit), so wait for the pipe to close before leaving
    try:
        sys.stdin.buffer.read()
    except Exception:
        pass
    _shm_cleanup()


_IDLE = []          # helper processes started ahead of time, waiting for a job on stdin


def _spawn():
    import subprocess
    env = dict(os.environ)
    pkg_parent = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = pkg_parent + os.pathsep + env.get("PYTHONPATH", "")
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env[k] = ""                               # host work only
    # torch.distributed.run pins OMP_NUM_THREADS to 1 for its workers: the helper's LAPACK calls get this rank's share of the cores
    # Without a cap OpenBLAS starts one thread per LOGICAL core of the host (256 on the GPU boxes, of which a job may use 16): the dense
    # Cholesky of the coarsest level then spends its time in thread management (0.75 s -> 0.2 s for 3 089 rows with 16 threads).
    world = max(1, int(env.get("WORLD_SIZE", "1") or 1))
    share = str(max(1, min(16, (os.cpu_count() or 1) // world)))
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        if world > 1 or k not in os.environ:
            env[k] = share
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.Popen([sys.executable, "-m", "knpemidg.setup_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)


def prestart(n=2):
    """Start up to n idle helpers now (Solver.__init__): they import their modules while the caller still builds its mesh.  Idle
    helpers that never get a job are killed at interpreter exit."""
    import atexit
    if os.environ.get("KNP_AMG_SERIAL_SETUP", "0") == "1":
        return
    if not _IDLE:
        atexit.register(_reap)
    while len(_IDLE) < n:
        try:
            _IDLE.append(_spawn())
        except OSError:
            break


def _reap():
    _shm_cleanup()
    while _IDLE:
        p = _IDLE.pop()
        try:
            p.kill()
            p.wait(timeout=5)
        except Exception:
            pass


def start(job):
    """Hand the job to a helper (one started ahead of time if there is one, else a new one) from a feeder thread (pipe I/O releases
    the GIL).  Returns a handle for collect()."""
    import threading
    proc = None
    while _IDLE and proc is None:
        cand = _IDLE.pop(0)
        if cand.poll() is None:
            proc = cand
    if proc is None:
        proc = _spawn()
    handle = {"proc": proc, "result": None}
    if os.path.isdir(_SHM_DIR):
        # the helper's result file is named (and, on every path, removed) by this process: a helper killed after it wrote the file
        # cannot clean up after itself
        job = dict(job, _result_file=_new_name())
        _shm_made.append(job["_result_file"])
        handle["result_file"] = job["_result_file"]
        handle["job_file"] = _new_name()                   # named here, so that it can be removed even if the hand-over itself is cut short
        import atexit
        atexit.register(_shm_cleanup)

    def feed():
        try:
            _dump(job, proc.stdin, name=handle.get("job_file"))
            handle["result"] = _load(proc.stdout)
            proc.stdin.close()                              # the helper waits for this before it removes what it wrote and leaves
        except BaseException as e:
            handle["result"] = {"error": "helper process: %r" % (e,)}
        finally:
            for pipe in (proc.stdin, proc.stdout):          # stdin first: the helper leaves when it sees it closed
                try:
                    pipe.close()
                except OSError:
                    pass
            proc.wait()
            _remove(handle.get("result_file"))              # already gone when the result was loaded; left behind by a helper that was killed
            _remove(handle.get("job_file"))                 # likewise: the helper unlinks it when it maps it
    th = threading.Thread(target=feed, name="knp-amg-helper-io", daemon=True)
    th.start()
    handle["thread"] = th
    return handle


def collect(handle):
    """Groups built by the helper, or None (with the reason on stderr) when it failed."""
    import time
    t0 = time.perf_counter()
    handle["thread"].join()
    handle["waited_s"] = time.perf_counter() - t0          # how long the caller stood still for the helper (0: it was done already)
    if os.environ.get("KNP_DEBUG"):
        print("[knpemidg] waited %.2f s for a hierarchy helper" % handle["waited_s"], file=sys.stderr)
    res = handle["result"] or {"error": "no result"}
    if "groups" in res:
        return res["groups"]
    print("[knpemidg] KNP hierarchy helper failed, building in-process: %s" % res.get("error", "?").splitlines()[0], file=sys.stderr)
    return None


def cancel(handle):
    try:
        handle["proc"].kill()
    except OSError:
        pass
    handle["thread"].join(timeout=5)


if __name__ == "__main__":
    main()
