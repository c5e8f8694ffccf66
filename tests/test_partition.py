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
