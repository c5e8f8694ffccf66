"""Multi-GPU path on the CPU: partition + halo tables validated with world_size-2/3 gloo runs in which every
rank applies the ORACLE operator to its owned+ghost sub-mesh and must reproduce the owned rows of the global
operator apply (DG needs exactly one facet-neighbour ghost layer)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emix_mesh():
    sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd"))
    sys.path.insert(0, os.path.join(ROOT, "examples", "emix_simulations"))
    from emix_common import load_mesh
    return load_mesh()


@pytest.mark.parametrize("case", ["idealized-slab", "idealized-rcb", "emix-rcb"])
def test_partition_tables_are_consistent(case):
    """Product host path of the multi-GPU run: partition -> local meshes -> halo tables, for x-slabs of the idealized BoxMesh
    and for the recursive coordinate bisection used on unstructured meshes (BASELINE configs[4], the EMIx reconstruction)."""
    sys.path.insert(0, os.path.join(ROOT, "knp-emi-dg_amd"))
    from knpemidg.mesh import make_mesh_3D
    from knpemidg.partition import Partition
    m, s, f = _emix_mesh() if case.startswith("emix") else make_mesh_3D(0, n_axons=1)
    method = case.split("-")[1]
    mtags = (1, 2) if case.startswith("emix") else (1,)
    for world in ((2, 3, 8) if method == "slab" else (2, 4, 8)):
        part = Partition(m, world, method=method)
        locs = [part.local(r) for r in range(world)]
        assert sum(l.nc_owned for l in locs) == m.num_cells()
        assert max(l.nc_owned for l in locs) - min(l.nc_owned for l in locs) <= (1 if method == "slab" else world)
        for r, l in enumerate(locs):
            assert len(l.peers) <= (2 if method == "slab" else world - 1)
            for q, sl, ro, rc in zip(l.peers, l.send_lists, l.recv_offsets, l.recv_counts):
                lq = locs[q]
                i = lq.peers.index(r)
                # what I send is what the peer expects to receive, in the same (global id) order
                assert np.array_equal(l.cells_global[sl], lq.cells_global[lq.recv_offsets[i]:lq.recv_offsets[i] + lq.recv_counts[i]])
                assert (sl < l.nc_owned).all() and ro >= l.nc_owned
            # every facet neighbour of an owned cell is present locally
            nb = l.mesh.facet_cells
            cf = l.mesh.cell_facets[:l.nc_owned]
            glob_int = (m.facet_cells[m.cell_facets[l.owned]][:, :, 1] >= 0)
            loc_int = (nb[cf][:, :, 1] >= 0)
            assert np.array_equal(glob_int, loc_int)
            sub_l, surf_l = l.localize(s, f, mtags)
            assert np.array_equal(sub_l.array(), s.array()[l.cells_global])
            # membrane facets kept == those touching an owned cell
            mem = np.nonzero(np.isin(surf_l.array(), mtags))[0]
            assert ((nb[mem, 0] < l.nc_owned) | (nb[mem, 1] < l.nc_owned)).all()
        # every membrane facet of the global mesh is kept by at least one rank (and by two when it lies on a cut)
        gmem = np.nonzero((m.facet_cells[:, 1] >= 0) & np.isin(f.array(), mtags))[0]
        seen = np.zeros(m.num_facets(), dtype=np.int64)
        for l in locs:
            _, surf_l = l.localize(s, f, mtags)
            seen[l.facets_global[np.isin(surf_l.array(), mtags)]] += 1
        assert (seen[gmem] >= 1).all() and seen.max() <= 2


def _worker(rank, world, port, q, method="slab"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    for p in (os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import knpemi_oracle as ko
    from common import synthetic_state, small_3d
    from knpemidg.partition import Partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, s, f = small_3d((12, 4, 4))
        pbg = ko.build_idealized(m, s.array(), f.array(), membrane_tags=(1,))
        x = synthetic_state(pbg)
        Ag, bg, _ = ko.assemble_emi(pbg, want_B=False)
        yg = (Ag @ x[0].ravel()).reshape(-1, pbg.nd)
        yk = [(ko.assemble_knp(pbg, k) @ x[k].ravel()).reshape(-1, pbg.nd) for k in range(pbg.N_ions)]
        bk = [ko.knp_rhs(pbg, k).reshape(-1, pbg.nd) for k in range(pbg.N_ions)]
        loc = Partition(m, world, method=method).local(rank)
        sub_l, surf_l = loc.localize(s, f, (1,))
        pbl = ko.build_idealized(loc.mesh, sub_l.array(), surf_l.array(), membrane_tags=(1,))
        cg = loc.cells_global
        no = loc.nc_owned

        def scatter(a):                       # owned values from the global array, ghosts via the halo tables
            out = np.zeros((len(cg),) + a.shape[1:])
            out[:no] = a[cg[:no]]
            return loc.exchange_host(out, dist)
        pbl.c = np.stack([scatter(pbg.c[k]) for k in range(pbg.N_ions)])
        pbl.c_prev_n = np.stack([scatter(pbg.c_prev_n[k]) for k in range(pbg.N_ions)])
        pbl.c_elim = scatter(pbg.c_elim)
        pbl.phi = scatter(pbg.phi)
        assert np.array_equal(pbl.phi, pbg.phi[cg])                       # ghosts arrived in the right slots
        pbl.phi_M = pbg.phi_M[loc.facets_global]
        for name in pbg.I_ch:
            pbl.I_ch[name] = pbg.I_ch[name][loc.facets_global]
        xl = scatter(x[0])
        Al, bl, _ = ko.assemble_emi(pbl, want_B=False)
        yl = (Al @ xl.ravel()).reshape(-1, pbl.nd)
        err = [np.abs(yl[:no] - yg[cg[:no]]).max() / np.abs(yg).max(),
               np.abs(bl.reshape(-1, pbl.nd)[:no] - bg.reshape(-1, pbg.nd)[cg[:no]]).max() / np.abs(bg).max()]
        for k in range(pbg.N_ions):
            ykl = (ko.assemble_knp(pbl, k) @ scatter(x[k]).ravel()).reshape(-1, pbl.nd)
            err.append(np.abs(ykl[:no] - yk[k][cg[:no]]).max() / np.abs(yk[k]).max())
            bkl = ko.knp_rhs(pbl, k).reshape(-1, pbl.nd)
            err.append(np.abs(bkl[:no] - bk[k][cg[:no]]).max() / np.abs(bk[k]).max())
        # global dot product = all-reduced owned partial sums
        import torch
        part = torch.tensor([float((xl[:no] * yl[:no]).sum())], dtype=torch.float64)
        dist.all_reduce(part)
        err.append(abs(part.item() - float((x[0] * yg).sum())) / abs(float((x[0] * yg).sum())))
        q.put((rank, max(err)))
    except Exception as e:                                                # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,method", [(2, "slab"), (3, "slab"), (4, "rcb"), (8, "rcb")])
def test_halo_exchange_gloo(world, method):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, method)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, e in res:
        assert not isinstance(e, str), e
        assert e < 1e-12, (rank, e)


@pytest.mark.parametrize("degree,world,method", [(1, 3, "slab"), (1, 4, "rcb"), (2, 2, "slab"), (2, 3, "rcb")])
def test_row_distributed_level0_algebra(degree, world, method, monkeypatch):
    """knpemidg/amg.py: Dist0Space (the row-distributed finest conforming level of a partitioned run, csrc/amg.hip dist0), emulated on
    the host with every rank's tables: the sub-assembled matrices add up to the global operator, the shared-dof tables of two peers list
    the same dofs in the same order, the rank-ordered accumulation gives every owner the same bits, and one level-0 visit of the V-cycle
    (smooth, residual, restriction, a replicated coarse map, prolongation, smooth) on the distributed data equals the global one."""
    from common import small_3d
    from knpemidg import amg
    from knpemidg.partition import Partition
    monkeypatch.setenv("KNP_AMG_MAXCOARSE", "60")         # a coarser level below the small mesh's finest one
    mesh, sub, surf = small_3d((8, 4, 4))
    cs = amg.ConformingSpace(mesh, surf.array(), [1])
    space = cs if degree == 1 else amg.ConformingSpaceP2(cs)
    nd = space.dof.shape[1]
    rng = np.random.default_rng(5)
    kappa = rng.uniform(0.5, 1.5, size=(mesh.num_cells(), nd))
    mem = np.nonzero((mesh.facet_cells[:, 1] >= 0) & np.isin(surf.array(), [1]))[0]
    assert len(mem) > 0
    levels = amg.build_emi_levels(cs, space if degree == 2 else None, surf.array(), [1], kappa, 2.0e2)
    if degree == 1:
        levels[0].cheb_degree = 2                     # exercise the extra Chebyshev step of the distributed smoother too
    assert len(levels) >= 2
    A = levels[0].A
    part = Partition(mesh, world, method=method)
    D = [amg.Dist0Space(space, part.owner, mem, r, world) for r in range(world)]
    Al = [d.local_matrix(kappa, membrane_C=2.0e2) for d in D]
    # (1) sub-assembled matrices add up to the global operator; every dof has an owner
    S = sum((sp_sel(d, space.n).T @ a @ sp_sel(d, space.n)) for d, a in zip(D, Al))
    assert abs(S - A).max() <= 1e-12 * abs(A).max()
    assert np.array_equal(np.unique(np.concatenate([d.verts for d in D])), np.arange(space.n))
    tabs = [d.interface_tables() for d in D]
    # (2) peers list the same shared dofs in the same order
    for r, (peers, lists, uvtx, aptr, asrc) in enumerate(tabs):
        for q, l in zip(peers, lists):
            pq, lq = tabs[q][0], tabs[q][1]
            assert r in pq
            assert np.array_equal(D[r].verts[l], D[q].verts[lq[pq.index(r)]])
        assert len(uvtx) == len(aptr) - 1 and aptr[-1] == len(asrc) and (asrc == -1).sum() == len(uvtx)

    def accumulate(vs):
        """what interface_accumulate does: pack per peer, 'receive' the peer's message, add in rank order"""
        send = [np.concatenate([v[l] for l in t[1]]) if t[0] else np.zeros(0) for v, t in zip(vs, tabs)]
        out = []
        for r, (v, (peers, lists, uvtx, aptr, asrc)) in enumerate(zip(vs, tabs)):
            recv = []
            for q, l in zip(peers, lists):
                pq, lq = tabs[q][0], tabs[q][1]
                k = pq.index(r)
                off = sum(len(x) for x in lq[:k])
                recv.append(send[q][off:off + len(lq[k])])
            recv = np.concatenate(recv) if recv else np.zeros(0)
            w = v.copy()
            for u in range(len(uvtx)):
                s_ = 0.0
                for k in range(aptr[u], aptr[u + 1]):
                    s_ += v[uvtx[u]] if asrc[k] < 0 else recv[asrc[k]]
                w[uvtx[u]] = s_
            out.append(w)
        return out

    # (3) accumulation: partial sums of a global vector -> every owner holds the global value, bit-identical between owners
    bg = rng.standard_normal(space.n)
    parts = [a @ np.ones(d.n) * 0 + (sp_sel(d, space.n) @ bg) / share_count(D, space.n)[d.verts] for d, a in zip(D, Al)]
    acc = accumulate(parts)
    full = np.zeros(space.n)
    seen = np.zeros(space.n, dtype=bool)
    for d, v in zip(D, acc):
        assert np.allclose(v, bg[d.verts], rtol=1e-13, atol=1e-14)
        assert np.array_equal(v[seen[d.verts]], full[d.verts][seen[d.verts]])
        full[d.verts] = v
        seen[d.verts] = True

    # (4) one level-0 visit: global reference
    g = levels[0]
    lmax, lmin = g.rho, g.cheb_lower * g.rho
    theta, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = theta / delta
    Cmap = rng.standard_normal((g.P.shape[1],)) * 1e-3               # a replicated (diagonal) stand-in for the coarser levels

    def smooth_global(x, b, zero):
        r = b.copy() if zero else b - A @ x
        d_ = g.dinv * r / theta
        x = d_.copy() if zero else x + d_
        rho = 1.0 / sigma
        for _ in range(1, g.cheb_degree):
            rho_new = 1.0 / (2.0 * sigma - rho)
            r = r - A @ d_
            d_ = rho_new * rho * d_ + 2.0 * rho_new / delta * g.dinv * r
            x = x + d_
            rho = rho_new
        return x
    b = A @ rng.standard_normal(space.n) + 0.1 * rng.standard_normal(space.n)
    xg = smooth_global(None, b, True)
    xg = xg + g.P @ (Cmap * (g.P.T @ (b - A @ xg)))
    xg = smooth_global(xg, b, False)

    # distributed: b as per-rank partial sums (what the DG restriction of the owned cells produces)
    loc = [d.localize(levels, a)[0] for d, a in zip(D, Al)]
    bp = [(sp_sel(d, space.n) @ b) / share_count(D, space.n)[d.verts] for d in D]

    def smooth_dist(xs, zero):
        if zero:
            rs = accumulate([p.copy() for p in bp])
        else:
            rs = accumulate([p - l.A @ x for p, l, x in zip(bp, loc, xs)])
        ds = [l.dinv * r / theta for l, r in zip(loc, rs)]
        xs = [d_.copy() for d_ in ds] if zero else [x + d_ for x, d_ in zip(xs, ds)]
        rho = 1.0 / sigma
        for _ in range(1, g.cheb_degree):
            rho_new = 1.0 / (2.0 * sigma - rho)
            ts = accumulate([l.A @ d_ for l, d_ in zip(loc, ds)])
            rs = [r - t for r, t in zip(rs, ts)]
            ds = [rho_new * rho * d_ + 2.0 * rho_new / delta * l.dinv * r for l, d_, r in zip(loc, ds, rs)]
            xs = [x + d_ for x, d_ in zip(xs, ds)]
            rho = rho_new
        return xs
    xs = smooth_dist(None, True)
    r1 = sum(l.R @ (p - l.A @ x) for l, p, x in zip(loc, bp, xs))     # the all-reduce of the level-1 right-hand side
    xs = [x + l.P @ (Cmap * r1) for l, x in zip(loc, xs)]
    xs = smooth_dist(xs, False)
    for d, x in zip(D, xs):
        assert np.allclose(x, xg[d.verts], rtol=1e-10, atol=1e-12 * np.abs(xg).max())
    # DG map: every DG dof of an owned cell finds its conforming dof among the rank's rows
    for r, d in enumerate(D):
        cells = np.nonzero(part.owner == r)[0]
        assert np.array_equal(d.verts[d.local_dg2cg(cells)], space.dof[cells])


def sp_sel(d, n):
    import scipy.sparse as sp
    return sp.csr_matrix((np.ones(d.n), (np.arange(d.n), d.verts)), shape=(d.n, n))


def share_count(D, n):
    cnt = np.zeros(n)
    for d in D:
        cnt[d.verts] += 1.0
    return cnt


def _dist0_worker(rank, world, port, q, method, degree):
    """One rank of test_row_distributed_level0_gloo: holds only ITS rows of the finest conforming level, exchanges the shared dofs with
    its peers through torch.distributed (gloo) and all-reduces the level-1 right-hand side -- the communication pattern of
    csrc/comm.hip interface_accumulate + csrc/amg.hip amg_vcycle_dist0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["KNP_AMG_MAXCOARSE"] = "60"
    for p in (os.path.join(ROOT, "knp-emi-dg_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from common import small_3d
    from knpemidg import amg
    from knpemidg.partition import Partition
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mesh, sub, surf = small_3d((max(8, 2 * world), 4, 4))
        cs = amg.ConformingSpace(mesh, surf.array(), [1])
        space = cs if degree == 1 else amg.ConformingSpaceP2(cs)
        rng = np.random.default_rng(5)                                     # same stream on every rank: global data, as in a partitioned run
        kappa = rng.uniform(0.5, 1.5, size=(mesh.num_cells(), space.dof.shape[1]))
        mem = np.nonzero((mesh.facet_cells[:, 1] >= 0) & np.isin(surf.array(), [1]))[0]
        levels = amg.build_emi_levels(cs, space if degree == 2 else None, surf.array(), [1], kappa, 2.0e2)
        g = levels[0]
        A = g.A
        part = Partition(mesh, world, method=method)
        d0 = amg.Dist0Space(space, part.owner, mem, rank, world)
        loc = d0.localize(levels, d0.local_matrix(kappa, membrane_C=2.0e2))[0]
        peers, lists, uvtx, aptr, asrc = d0.interface_tables()
        off = np.concatenate([[0], np.cumsum([len(l) for l in lists])]).astype(int)

        def accumulate(v):
            send = [torch.from_numpy(np.ascontiguousarray(v[l])) for l in lists]
            recv = [torch.empty(len(l), dtype=torch.float64) for l in lists]
            reqs = [dist.isend(s_, p_) for s_, p_ in zip(send, peers)] + [dist.irecv(r_, p_) for r_, p_ in zip(recv, peers)]
            for rq in reqs:
                rq.wait()
            rbuf = np.concatenate([r_.numpy() for r_ in recv]) if recv else np.zeros(0)
            w = v.copy()
            for u in range(len(uvtx)):
                s_ = 0.0
                for k in range(aptr[u], aptr[u + 1]):
                    s_ += v[uvtx[u]] if asrc[k] < 0 else rbuf[asrc[k]]
                w[uvtx[u]] = s_
            return w
        assert off[-1] == sum(len(l) for l in lists)
        lmax, lmin = g.rho, g.cheb_lower * g.rho
        theta = 0.5 * (lmax + lmin)
        b = A @ rng.standard_normal(space.n) + 0.1 * rng.standard_normal(space.n)
        Cmap = rng.standard_normal(g.P.shape[1]) * 1e-3
        # global reference (one Chebyshev step per smoothing: the shipped level-0 smoother of the P1 hierarchies)
        xg = g.dinv * b / theta
        xg = xg + g.P @ (Cmap * (g.P.T @ (b - A @ xg)))
        xg = xg + g.dinv * (b - A @ xg) / theta
        # this rank's share of b: full values on dofs it alone holds, an equal split on shared ones (what matters is that the parts add up)
        cnt = np.zeros(space.n)
        for vq in d0.rank_verts:
            cnt[vq] += 1.0
        bp = b[d0.verts] / cnt[d0.verts]
        x = loc.dinv * accumulate(bp) / theta
        r1 = torch.from_numpy(loc.R @ (bp - loc.A @ x))
        dist.all_reduce(r1)
        x = x + loc.P @ (Cmap * r1.numpy())
        x = x + loc.dinv * accumulate(bp - loc.A @ x) / theta
        q.put((rank, float(np.abs(x - xg[d0.verts]).max() / np.abs(xg).max())))
    except Exception:                                                     # pragma: no cover
        import traceback
        q.put((rank, "ERR " + traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,method,degree", [(2, "slab", 1), (3, "rcb", 1), (2, "slab", 2), (4, "slab", 1), (8, "rcb", 1), (4, "rcb", 2)])
def test_row_distributed_level0_gloo(world, method, degree):
    """The row-distributed finest conforming level with one PROCESS per rank: every rank builds only its own rows and tables
    (knpemidg.amg.Dist0Space), the shared dofs travel point to point and the level-1 right-hand side through an all-reduce (gloo); one
    level-0 visit of the V-cycle equals the global one on every rank's rows."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world + 10 * degree
    procs = [ctx.Process(target=_dist0_worker, args=(r, world, port, q, method, degree)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, e in res:
        assert not isinstance(e, str), e
        assert e < 1e-10, (rank, e)
