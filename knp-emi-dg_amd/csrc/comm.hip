// Multi-GPU plumbing: one process per GPU, RCCL over xGMI.
//  * halo exchange of facet-neighbour (ghost) cell DoFs: packed per peer, grouped ncclSend/ncclRecv
//    (messages are ~80 KB per field for slab partitions -> latency bound; one group per exchange,
//    every peer is one xGMI hop);
//  * Krylov inner products: one ncclAllReduce of the packed partial sums per reduction point.
// Replaces DOLFIN ghost updates + PETSc VecGhost / MatMult scatters + KSP reductions
// (reference: src/knpemidg/solver.py:16, 529, 789).
#include "../../include/knpemi_hip.h"
#include "knpemi_internal.hpp"
#include <rccl/rccl.h>
#include <cstring>
#include <cstdlib>

#define NCCLCHK(ctx, call)                                                         \
    do {                                                                           \
        ncclResult_t r_ = (call);                                                  \
        if (r_ != ncclSuccess) {                                                   \
            (ctx)->err = std::string(#call) + ": " + ncclGetErrorString(r_);       \
            return -6;                                                             \
        }                                                                          \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= 128, "unique id must fit the ABI buffer");

void comm_destroy(knp_ctx* c) {
    if (c->comm_halo) { ncclCommDestroy((ncclComm_t)c->comm_halo); c->comm_halo = nullptr; }
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c->halo_stream) { hipStreamDestroy(c->halo_stream); c->halo_stream = nullptr; }
    if (c->halo_ready) { hipEventDestroy(c->halo_ready); c->halo_ready = nullptr; }
    if (c->halo_done) { hipEventDestroy(c->halo_done); c->halo_done = nullptr; }
}

int allreduce_red(knp_ctx* c, double* red, int count) {
    if (!c->comm) { c->err = "allreduce without communicator"; return -6; }
    NCCLCHK(c, ncclAllReduce(red, red, count, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
    return 0;
}

int allreduce_max(knp_ctx* c, double* host_value) {
    if (!c->comm) { c->err = "allreduce without communicator"; return -6; }
    double* d = c->scal + KNP_MAX_SYS * 12;            // reduction scratch (krylov.hpp: KS_N = 12)
    HIPCHK(c, hipMemcpyAsync(d, host_value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclAllReduce(d, d, 1, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(host_value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// out layout for one peer: [field][cell][nv]
__global__ void k_halo_pack(const double* __restrict__ v, const int32_t* __restrict__ idx, int64_t cnt, int nfields,
                            int64_t field_stride, int nv, double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= cnt * nv) return;
    const int64_t i = t / nv;
    const int a = (int)(t % nv);
    const int64_t cell = idx[i];
    for (int f = 0; f < nfields; ++f) out[((int64_t)f * cnt + i) * nv + a] = v[(int64_t)f * field_stride + cell * nv + a];
}

static int halo_exchange_on(knp_ctx* c, double* v, int nfields, hipStream_t st, ncclComm_t comm);

int halo_exchange(knp_ctx* c, double* v, int nfields) {
    if (!c->dist || c->halo_peer.empty()) return 0;
    if (!c->comm) { c->err = "halo exchange without communicator"; return -6; }
    return halo_exchange_on(c, v, nfields, c->stream, (ncclComm_t)c->comm);
}

// Operator apply with its halo exchange.  The owned cells are ordered interior-first (knp_set_interior), so the exchange of the
// input vector (pack kernel + grouped ncclSend/ncclRecv on the halo stream, through a communicator of its own so that it can
// run next to an all-reduce of the solver's stream) overlaps the interior launch; the boundary launch waits for it.
// Per apply and rank: 2 messages per peer and field (one each way), cells_on_the_cut x nd x 8 B each.
int launch_emi_apply(knp_ctx* c, const double* x, const double* kappa, double* y);
int launch_knp_apply(knp_ctx* c, const double* x, const double* dnphi, double* y);

int dist_apply(knp_ctx* c, int which, double* x, const double* coef, double* y) {
    auto launch = [&]() { return which == 0 ? launch_emi_apply(c, x, coef, y) : launch_knp_apply(c, x, coef, y); };
    if (!c->dist) {
        // KNP_FORCE_SPLIT=1 (tests): interior and boundary launches without a communicator, ghost values as uploaded
        const char* fs = getenv("KNP_FORCE_SPLIT");
        if (!(fs && atoi(fs) == 1) || c->m.n_interior >= c->m.nc_owned) return launch();
        const int64_t n_own = c->m.nc_owned;
        c->m.c_begin = 0; c->m.c_end = c->m.n_interior;
        int rc = launch();
        c->m.c_begin = c->m.n_interior; c->m.c_end = n_own;
        if (!rc) rc = launch();
        c->m.c_begin = 0; c->m.c_end = n_own;
        return rc;
    }
    const int nfields = which == 0 ? 1 : c->p.n_sys;
    int rc;
    if (!c->comm_halo || c->halo_peer.empty() || c->m.n_interior <= 0) {
        if ((rc = halo_exchange(c, x, nfields))) return rc;
        return launch();
    }
    HIPCHK(c, hipEventRecord(c->halo_ready, c->stream));                 // x is final
    HIPCHK(c, hipStreamWaitEvent(c->halo_stream, c->halo_ready, 0));
    if ((rc = halo_exchange_on(c, x, nfields, c->halo_stream, (ncclComm_t)c->comm_halo))) return rc;
    HIPCHK(c, hipEventRecord(c->halo_done, c->halo_stream));
    const int64_t n_own = c->m.nc_owned;
    c->m.c_begin = 0; c->m.c_end = c->m.n_interior;
    rc = launch();                                                       // needs no ghost value
    if (!rc) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->halo_done, 0));
        c->m.c_begin = c->m.n_interior; c->m.c_end = n_own;
        rc = launch();
    }
    c->m.c_begin = 0; c->m.c_end = n_own;
    return rc;
}

static int halo_exchange_on(knp_ctx* c, double* v, int nfields, hipStream_t st, ncclComm_t comm) {
    if (nfields > KNP_MAX_SYS) { c->err = "halo exchange: too many fields"; return -1; }
    const int NV = c->nd;
    const int64_t stride = c->m.nc * NV;
    const int np = (int)c->halo_peer.size();
    for (int p = 0; p < np; ++p) {
        const int64_t cnt = c->halo_send_cnt[p];
        if (!cnt) continue;
        const int64_t n = cnt * NV;
        hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)v,
                           (const int32_t*)(c->halo_send_idx + c->halo_send_off[p]), cnt, nfields, stride, NV,
                           c->halo_sendbuf + c->halo_send_off[p] * KNP_MAX_SYS * NV);
    }
    HIPCHK(c, hipGetLastError());
    NCCLCHK(c, ncclGroupStart());
    for (int p = 0; p < np; ++p) {
        const int peer = c->halo_peer[p];
        // one message per field, matching the receiver's per-field ncclRecv calls one-to-one (same order)
        for (int f = 0; f < nfields && c->halo_send_cnt[p]; ++f)
            NCCLCHK(c, ncclSend(c->halo_sendbuf + c->halo_send_off[p] * KNP_MAX_SYS * NV + (int64_t)f * c->halo_send_cnt[p] * NV,
                                (size_t)(c->halo_send_cnt[p] * NV), ncclDouble, peer, comm, st));
        for (int f = 0; f < nfields && c->halo_recv_cnt[p]; ++f)
            NCCLCHK(c, ncclRecv(v + (int64_t)f * stride + c->halo_recv_off[p] * NV, (size_t)(c->halo_recv_cnt[p] * NV), ncclDouble,
                                peer, comm, st));
    }
    NCCLCHK(c, ncclGroupEnd());
    return 0;
}

extern "C" {

int knp_comm_unique_id(char* out128) {
    if (!out128) return -1;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return -6;
    memset(out128, 0, 128);
    memcpy(out128, &id, sizeof(id));
    return 0;
}

int knp_comm_init(knp_ctx* c, int rank, int nranks, const char* id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
    c->rank = rank;
    c->nranks = nranks;
    // KNP_FORCE_COMM=1: build the communicator and take every collective code path even with one rank (a 1-GPU box can
    // then exercise ncclCommInitRank / ncclAllReduce on the solver's stream; RCCL refuses two ranks on one device)
    if (nranks == 1 && !(getenv("KNP_FORCE_COMM") && atoi(getenv("KNP_FORCE_COMM")) == 1)) return 0;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    HIPCHK(c, hipSetDevice(c->device));
    ncclComm_t comm;
    NCCLCHK(c, ncclCommInitRank(&comm, nranks, id, rank));
    c->comm = comm;
    c->dist = true;
    return 0;
}

// Second communicator for the overlapped halo exchanges (optional: without it every exchange runs on the context's stream in
// front of the apply).  Same rank / size as knp_comm_init; its own unique id.
int knp_comm_init_halo(knp_ctx* c, const char* id128) {
    if (!c || !id128) return -1;
    if (!c->dist) return 0;
    if (getenv("KNP_HALO_OVERLAP") && atoi(getenv("KNP_HALO_OVERLAP")) == 0) return 0;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    HIPCHK(c, hipSetDevice(c->device));
    ncclComm_t comm;
    NCCLCHK(c, ncclCommInitRank(&comm, c->nranks, id, c->rank));
    c->comm_halo = comm;
    HIPCHK(c, hipStreamCreate(&c->halo_stream));
    HIPCHK(c, hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->halo_done, hipEventDisableTiming));
    return 0;
}

int knp_halo_tables(knp_ctx* c, int npeers, const int32_t* peers, const int64_t* send_counts, const int32_t* send_cells,
                    const int64_t* recv_offsets, const int64_t* recv_counts) {
    if (!c || npeers < 0) return -1;
    c->halo_peer.clear(); c->halo_send_off.clear(); c->halo_send_cnt.clear(); c->halo_recv_off.clear(); c->halo_recv_cnt.clear();
    int64_t total = 0;
    for (int p = 0; p < npeers; ++p) {
        if (peers[p] < 0 || peers[p] >= c->nranks || peers[p] == c->rank) { c->err = "halo_tables: bad peer rank"; return -1; }
        if (recv_offsets[p] < c->m.nc_owned || recv_offsets[p] + recv_counts[p] > c->m.nc) {
            c->err = "halo_tables: ghost range outside [nc_owned, nc)"; return -1;
        }
        c->halo_peer.push_back(peers[p]);
        c->halo_send_off.push_back(total);
        c->halo_send_cnt.push_back(send_counts[p]);
        c->halo_recv_off.push_back(recv_offsets[p]);
        c->halo_recv_cnt.push_back(recv_counts[p]);
        total += send_counts[p];
    }
    for (int64_t i = 0; i < total; ++i)
        if (send_cells[i] < 0 || send_cells[i] >= c->m.nc_owned) { c->err = "halo_tables: send cell is not owned"; return -1; }
    c->halo_send_total = total;
    hipFree(c->halo_send_idx); c->halo_send_idx = nullptr;
    hipFree(c->halo_sendbuf); c->halo_sendbuf = nullptr;
    if (total) {
        HIPCHK(c, hipMalloc((void**)&c->halo_send_idx, sizeof(int32_t) * total));
        HIPCHK(c, hipMemcpy(c->halo_send_idx, send_cells, sizeof(int32_t) * total, hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->halo_sendbuf, sizeof(double) * total * KNP_MAX_SYS * c->nd));
    }
    return 0;
}

}  // extern "C"
