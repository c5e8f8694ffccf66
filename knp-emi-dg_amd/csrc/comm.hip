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
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#define NCCLCHK(ctx, call)                                                         \
    do {                                                                           \
        ncclResult_t r_ = (call);                                                  \
        if (r_ != ncclSuccess) {                                                   \
            (ctx)->err = std::string(#call) + ": " + ncclGetErrorString(r_);       \
            return -6;                                                             \
        }                                                                          \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= 128, "unique id must fit the ABI buffer");

static void shm_destroy(knp_ctx* c);

void comm_destroy(knp_ctx* c) {
    shm_destroy(c);
    if (c->comm_halo) { ncclCommDestroy((ncclComm_t)c->comm_halo); c->comm_halo = nullptr; }
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    if (c->halo_stream) { hipStreamDestroy(c->halo_stream); c->halo_stream = nullptr; }
    if (c->halo_ready) { hipEventDestroy(c->halo_ready); c->halo_ready = nullptr; }
    if (c->halo_done) { hipEventDestroy(c->halo_done); c->halo_done = nullptr; }
    hipFree(c->if_idx); hipFree(c->if_uvtx); hipFree(c->if_aptr); hipFree(c->if_asrc); hipFree(c->if_send); hipFree(c->if_recv);
    c->if_idx = c->if_uvtx = c->if_aptr = c->if_asrc = nullptr;
    c->if_send = c->if_recv = nullptr;
}

static int shm_allreduce(knp_ctx* c, double* dev, int count, bool is_max);
static int shm_halo_exchange(knp_ctx* c, double* v, int nfields, hipStream_t st);

int allreduce_red(knp_ctx* c, double* red, int count) {
    if (c->shm) return shm_allreduce(c, red, count, false);
    if (!c->comm) { c->err = "allreduce without communicator"; return -6; }
    NCCLCHK(c, ncclAllReduce(red, red, count, ncclDouble, ncclSum, (ncclComm_t)c->comm, c->stream));
    return 0;
}

int allreduce_max(knp_ctx* c, double* host_value) {
    double* d = c->scal + KNP_MAX_SYS * 12;            // reduction scratch (krylov.hpp: KS_N = 12)
    if (c->shm) {
        HIPCHK(c, hipMemcpyAsync(d, host_value, sizeof(double), hipMemcpyHostToDevice, c->stream));
        int rc = shm_allreduce(c, d, 1, true);
        if (rc) return rc;
        HIPCHK(c, hipMemcpy(host_value, d, sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }
    if (!c->comm) { c->err = "allreduce without communicator"; return -6; }
    HIPCHK(c, hipMemcpyAsync(d, host_value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclAllReduce(d, d, 1, ncclDouble, ncclMax, (ncclComm_t)c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(host_value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// max over the ranks of a status word holding the bits of a non-negative float (they order like ints): the cell Peclet number of
// knp_update_dnphi, so that every rank of a solve picks the same block-Jacobi data.  RCCL: enqueued on the solver's stream, no host
// synchronisation; shm (validation transport): staged through the host.
int allreduce_max_word(knp_ctx* c, int* dev_word) {
    if (c->shm) {
        int bits = 0;
        HIPCHK(c, hipMemcpyAsync(&bits, dev_word, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        float pe;
        memcpy(&pe, &bits, sizeof(pe));
        double v = pe;
        int rc = allreduce_max(c, &v);
        if (rc) return rc;
        pe = (float)v;
        memcpy(&bits, &pe, sizeof(pe));
        HIPCHK(c, hipMemcpy(dev_word, &bits, sizeof(int), hipMemcpyHostToDevice));
        return 0;
    }
    if (!c->comm) { c->err = "allreduce without communicator"; return -6; }
    NCCLCHK(c, ncclAllReduce(dev_word, dev_word, 1, ncclInt32, ncclMax, (ncclComm_t)c->comm, c->stream));
    return 0;
}

// sum over the ranks of n host values (n <= KNP_MAX_SYS * KNP_MAX_RED = 56), identical bits on every rank
int allreduce_sum_host(knp_ctx* c, double* host_values, int n) {
    if (n < 0 || n > KNP_MAX_SYS * KNP_MAX_RED) { c->err = "allreduce_sum_host: at most 56 values"; return -1; }
    if (!c->dist || n == 0) return 0;
    double* d = c->scal + KNP_MAX_SYS * 12;            // reduction scratch: KNP_MAX_SYS * KNP_MAX_RED doubles (krylov.hip)
    HIPCHK(c, hipMemcpyAsync(d, host_values, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    int rc = allreduce_red(c, d, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(host_values, d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
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

// ---------------------------------------------------------------------------------------------------------------------------
// Host-staged communicator through POSIX shared memory (knp_comm_init_shm): the same three primitives as the RCCL path -- summed /
// maximised reductions and the peer halo exchange -- for ranks that are PROCESSES OF ONE NODE, possibly sharing one GPU (RCCL
// refuses two ranks on a device).  Its purpose is validation: it runs the whole partitioned solver (partition, ghost layer, halo
// tables, pack / unpack, all-reduced Krylov scalars and restricted residuals, replicated hierarchies, membrane facets on a cut) with
// 2..4 ranks on a one-GPU box against the single-rank solution (tests/test_gpu_multirank.py).  Every transfer is staged through the
// host and every step ends in a barrier, so it is slow by construction and never the measured path.
// Segment layout: header | per rank { 2 directories[SHM_MAX_PEERS] | reduce slot [red_cap doubles] | outbox [out_cap doubles] }.
// ---------------------------------------------------------------------------------------------------------------------------
#define SHM_MAX_PEERS 64
struct ShmHeader {
    std::atomic<uint32_t> arrive;
    std::atomic<uint32_t> gen;
    uint32_t nranks;
    std::atomic<uint32_t> fail;       // raised by a rank that cannot take part in an exchange; read by every rank behind the next barrier
    uint64_t red_cap, out_cap;
};
struct ShmDirEntry { int64_t peer, off, cnt; };
struct ShmComm {
    int rank = 0, nranks = 1;
    size_t bytes = 0, rank_stride = 0;
    char* base = nullptr;
    uint64_t red_cap = 0, out_cap = 0;
    std::vector<double> tmp;
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    char* rank_base(int r) const { return base + 4096 + (size_t)r * rank_stride; }
    ShmDirEntry* dir(int r) const { return reinterpret_cast<ShmDirEntry*>(rank_base(r)); }                   // cell halo messages
    ShmDirEntry* dir_if(int r) const { return dir(r) + SHM_MAX_PEERS; }                                      // interface-dof messages
    double* red(int r) const { return reinterpret_cast<double*>(rank_base(r) + sizeof(ShmDirEntry) * 2 * SHM_MAX_PEERS); }
    double* out(int r) const { return red(r) + red_cap; }
};

static int shm_barrier(knp_ctx* c, ShmComm* s) {
    ShmHeader* h = s->hdr();
    const uint32_t g = h->gen.load(std::memory_order_acquire);
    if (h->arrive.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)s->nranks) {
        h->arrive.store(0, std::memory_order_relaxed);
        h->gen.store(g + 1, std::memory_order_release);
        return 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->gen.load(std::memory_order_acquire) == g) {
        sched_yield();
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { c->err = "shm communicator: a rank did not reach the barrier"; return -6; }
    }
    return 0;
}

static int shm_allreduce(knp_ctx* c, double* dev, int count, bool is_max) {
    ShmComm* s = (ShmComm*)c->shm;
    if ((uint64_t)count > s->red_cap) { c->err = "shm communicator: reduction longer than the slot (KNP_SHM_RED_DOUBLES)"; return -6; }
    HIPCHK(c, hipMemcpyAsync(s->red(s->rank), dev, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc;
    if ((rc = shm_barrier(c, s))) return rc;
    s->tmp.assign(s->red(0), s->red(0) + count);                      // rank order: every rank forms the same sum bit for bit
    for (int r = 1; r < s->nranks; ++r) {
        const double* q = s->red(r);
        if (is_max) for (int i = 0; i < count; ++i) s->tmp[i] = q[i] > s->tmp[i] ? q[i] : s->tmp[i];
        else for (int i = 0; i < count; ++i) s->tmp[i] += q[i];
    }
    if ((rc = shm_barrier(c, s))) return rc;                          // nobody refills a slot that is still being read
    HIPCHK(c, hipMemcpyAsync(dev, s->tmp.data(), sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int shm_halo_exchange(knp_ctx* c, double* v, int nfields, hipStream_t st) {
    ShmComm* s = (ShmComm*)c->shm;
    const int NV = c->nd;
    const int64_t stride = c->m.nc * NV;
    const int np = (int)c->halo_peer.size();
    const bool fits = (uint64_t)(c->halo_send_total * KNP_MAX_SYS * NV) <= s->out_cap;
    if (!fits) s->hdr()->fail.store(1, std::memory_order_release);
    for (int p = 0; p < np && fits; ++p) {
        const int64_t cnt = c->halo_send_cnt[p];
        if (!cnt) continue;
        const int64_t n = cnt * NV;
        double* seg = c->halo_sendbuf + c->halo_send_off[p] * KNP_MAX_SYS * NV;
        hipLaunchKernelGGL(k_halo_pack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)v,
                           (const int32_t*)(c->halo_send_idx + c->halo_send_off[p]), cnt, nfields, stride, NV, seg);
        HIPCHK(c, hipMemcpyAsync(s->out(s->rank) + c->halo_send_off[p] * KNP_MAX_SYS * NV, seg, sizeof(double) * nfields * n,
                                 hipMemcpyDeviceToHost, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    int rc;
    if ((rc = shm_barrier(c, s))) return rc;
    if (s->hdr()->fail.load(std::memory_order_acquire)) { c->err = "shm communicator: outbox too small on a rank (KNP_SHM_OUT_DOUBLES)"; return -6; }
    for (int p = 0; p < np; ++p) {
        const int peer = c->halo_peer[p];
        const int64_t want = c->halo_recv_cnt[p];
        if (!want) continue;
        const ShmDirEntry* d = s->dir(peer);
        int64_t off = -1;
        for (int k = 0; k < SHM_MAX_PEERS; ++k)
            if (d[k].peer == s->rank && d[k].cnt > 0) { if (d[k].cnt != want) { c->err = "shm halo exchange: count mismatch with the peer"; return -6; } off = d[k].off; break; }
        if (off < 0) { c->err = "shm halo exchange: the peer sends nothing to this rank"; return -6; }
        const double* src = s->out(peer) + off * KNP_MAX_SYS * NV;      // [field][cell][nv] of the peer's message
        for (int f = 0; f < nfields; ++f)
            HIPCHK(c, hipMemcpyAsync(v + (int64_t)f * stride + c->halo_recv_off[p] * NV, src + (int64_t)f * want * NV, sizeof(double) * want * NV,
                                     hipMemcpyHostToDevice, st));
    }
    HIPCHK(c, hipStreamSynchronize(st));
    return shm_barrier(c, s);                                          // outboxes may be refilled
}

// ---------------------------------------------------------------------------------------------------------------------------
// Interface exchange of the row-distributed conforming level (amg.hip: dist0).  Message p carries this rank's partial sums at the dofs
// it shares with peer p ([position][column]); both sides list those dofs in ascending global order, so what arrives from p has the
// layout of what was sent to p.  k_if_add then forms, per shared dof, the sum over ALL its owners in ascending rank order -- every
// owner gets the same bits.  Messages: r=2 mesh, 8 slabs: ~3 k dofs x 8 B x columns per peer, i.e. latency only.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void k_if_pack(const double* __restrict__ v, const int32_t* __restrict__ idx, int64_t total, int64_t n, int ncol, int nil,
                          double* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total * ncol) return;
    const int64_t i = t / ncol;
    const int j = (int)(t % ncol);
    out[t] = v[((int64_t)(j / nil) * n + idx[i]) * nil + (j % nil)];
}

__global__ void k_if_add(double* __restrict__ v, const int32_t* __restrict__ uvtx, const int32_t* __restrict__ aptr,
                         const int32_t* __restrict__ asrc, const double* __restrict__ recv, int64_t nuniq, int64_t n, int ncol, int nil) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nuniq * ncol) return;
    const int64_t u = t / ncol;
    const int j = (int)(t % ncol);
    const int64_t q = ((int64_t)(j / nil) * n + uvtx[u]) * nil + (j % nil);
    const double own = v[q];
    double s = 0.0;
    for (int k = aptr[u]; k < aptr[u + 1]; ++k) {
        const int src = asrc[k];
        s += src < 0 ? own : recv[(int64_t)src * ncol + j];
    }
    v[q] = s;
}

static int shm_interface_exchange(knp_ctx* c, int ncol) {
    ShmComm* s = (ShmComm*)c->shm;
    // a rank whose outbox is too small must not leave before the barrier its peers wait in: it raises the segment's fail word and
    // every rank returns the error behind the barrier
    const bool fits = (uint64_t)(c->if_total * ncol) <= s->out_cap;
    if (!fits) s->hdr()->fail.store(1, std::memory_order_release);
    else if (c->if_total)
        HIPCHK(c, hipMemcpyAsync(s->out(s->rank), c->if_send, sizeof(double) * c->if_total * ncol, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int rc;
    if ((rc = shm_barrier(c, s))) return rc;
    if (s->hdr()->fail.load(std::memory_order_acquire)) {
        c->err = fits ? "shm communicator: a peer's outbox is too small for the interface exchange"
                      : "shm communicator: outbox too small for the interface exchange (KNP_SHM_OUT_DOUBLES)";
        return -6;
    }
    for (size_t p = 0; p < c->if_peer.size(); ++p) {
        const int peer = c->if_peer[p];
        const ShmDirEntry* d = s->dir_if(peer);
        int64_t off = -1;
        for (int k = 0; k < SHM_MAX_PEERS; ++k)
            if (d[k].peer == s->rank && d[k].cnt > 0) {
                if (d[k].cnt != c->if_cnt[p]) { c->err = "shm interface exchange: the peer shares a different number of dofs"; return -6; }
                off = d[k].off; break;
            }
        if (off < 0) { c->err = "shm interface exchange: the peer lists no dofs shared with this rank"; return -6; }
        HIPCHK(c, hipMemcpyAsync(c->if_recv + c->if_off[p] * ncol, s->out(peer) + off * ncol, sizeof(double) * c->if_cnt[p] * ncol,
                                 hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return shm_barrier(c, s);                                          // outboxes may be refilled
}

int interface_accumulate(knp_ctx* c, double* v, int64_t n, int ncol) {
    if (!c->dist) return 0;
    if (ncol < 1 || ncol > KNP_MAX_SYS) { c->err = "interface exchange: bad column count"; return -1; }
    const int nil = (ncol % 2 == 0) ? 2 : 1;
    if (c->if_total)
        hipLaunchKernelGGL(k_if_pack, dim3((unsigned)((c->if_total * ncol + 255) / 256)), dim3(256), 0, c->stream, (const double*)v,
                           (const int32_t*)c->if_idx, c->if_total, n, ncol, nil, c->if_send);
    if (c->shm) {                                                      // every rank takes part in the barriers, with or without peers
        int rc = shm_interface_exchange(c, ncol);
        if (rc) return rc;
    } else if (!c->if_peer.empty()) {
        if (!c->comm) { c->err = "interface exchange without communicator"; return -6; }
        NCCLCHK(c, ncclGroupStart());
        for (size_t p = 0; p < c->if_peer.size(); ++p) {
            NCCLCHK(c, ncclSend(c->if_send + c->if_off[p] * ncol, (size_t)(c->if_cnt[p] * ncol), ncclDouble, c->if_peer[p], (ncclComm_t)c->comm, c->stream));
            NCCLCHK(c, ncclRecv(c->if_recv + c->if_off[p] * ncol, (size_t)(c->if_cnt[p] * ncol), ncclDouble, c->if_peer[p], (ncclComm_t)c->comm, c->stream));
        }
        NCCLCHK(c, ncclGroupEnd());
    }
    if (c->if_nuniq)
        hipLaunchKernelGGL(k_if_add, dim3((unsigned)((c->if_nuniq * ncol + 255) / 256)), dim3(256), 0, c->stream, v, (const int32_t*)c->if_uvtx,
                           (const int32_t*)c->if_aptr, (const int32_t*)c->if_asrc, (const double*)c->if_recv, c->if_nuniq, n, ncol, nil);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static void shm_destroy(knp_ctx* c) {
    ShmComm* s = (ShmComm*)c->shm;
    if (!s) return;
    if (s->base) munmap(s->base, s->bytes);
    delete s;
    c->shm = nullptr;
}

int halo_exchange(knp_ctx* c, double* v, int nfields) {
    if (!c->dist) return 0;
    if (c->shm) {
        if (nfields > KNP_MAX_SYS) { c->err = "halo exchange: too many fields"; return -1; }
        return shm_halo_exchange(c, v, nfields, c->stream);          // every rank takes part in the barriers, with or without peers
    }
    if (c->halo_peer.empty()) return 0;
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
    // overlap needs a second channel for the exchange: the halo communicator (RCCL) or the shm communicator's own stream.
    // (shm: every rank takes the same branch -- the barriers inside the exchange must pair up -- so the peer list does not decide)
    // The choice of channel must not depend on anything rank-local: a rank whose owned cells ALL touch the cut (n_interior == 0)
    // still posts its sends / receives on the halo communicator, like its peers -- point-to-point calls on different
    // communicators never pair.  Its interior launch is empty (the dispatchers return at c_end <= c_begin).
    const bool overlap = c->halo_stream && (c->shm || (c->comm_halo && !c->halo_peer.empty()));
    if (!overlap) {
        if ((rc = halo_exchange(c, x, nfields))) return rc;
        return launch();
    }
    HIPCHK(c, hipEventRecord(c->halo_ready, c->stream));                 // x is final
    HIPCHK(c, hipStreamWaitEvent(c->halo_stream, c->halo_ready, 0));
    const int64_t n_own = c->m.nc_owned;
    c->m.c_begin = 0; c->m.c_end = c->m.n_interior;
    rc = launch();                                                       // needs no ghost value: runs while the exchange is in flight
    c->m.c_begin = 0; c->m.c_end = n_own;
    if (rc) return rc;
    if (c->shm) rc = shm_halo_exchange(c, x, nfields, c->halo_stream);   // (host-staged: returns when the ghosts have arrived)
    else rc = halo_exchange_on(c, x, nfields, c->halo_stream, (ncclComm_t)c->comm_halo);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->halo_done, c->halo_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->halo_done, 0));
    c->m.c_begin = c->m.n_interior; c->m.c_end = n_own;
    rc = launch();
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

// The halo stream carries the pack kernel and RCCL's send / receive kernels while the interior launch of the apply occupies the
// chip: highest priority, so that their workgroups are dispatched ahead of the interior launch's queued ones (the interior launch
// also leaves a few CUs unoccupied when a communicator is active: apply_p1.hip halo_grid).
static hipError_t create_halo_stream(knp_ctx* c) {
    int lo = 0, hi = 0;                                                     // numerically LOWER = higher priority
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
    return hipStreamCreateWithPriority(&c->halo_stream, hipStreamNonBlocking, hi);
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
    HIPCHK(c, create_halo_stream(c));
    HIPCHK(c, hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->halo_done, hipEventDisableTiming));
    return 0;
}

// Host-staged communicator over a POSIX shared-memory segment `name` (all ranks of one node; see the block comment above).
// red_doubles / out_doubles: capacity of a rank's reduction slot and halo outbox.  Rank 0 creates the segment, the others attach.
int knp_comm_init_shm(knp_ctx* c, int rank, int nranks, const char* name, int64_t red_doubles, int64_t out_doubles) {
    if (!c || !name || nranks < 1 || rank < 0 || rank >= nranks || red_doubles < 64 || out_doubles < 0) return -1;
    if (c->comm || c->shm) { c->err = "communicator already initialised"; return -1; }
    ShmComm* s = new ShmComm();
    s->rank = rank; s->nranks = nranks; s->red_cap = (uint64_t)red_doubles; s->out_cap = (uint64_t)out_doubles;
    s->rank_stride = ((sizeof(ShmDirEntry) * 2 * SHM_MAX_PEERS + sizeof(double) * (s->red_cap + s->out_cap) + 4095) / 4096) * 4096;
    s->bytes = 4096 + s->rank_stride * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)s->bytes) != 0) { c->err = "shm communicator: cannot create the segment"; if (fd >= 0) close(fd); delete s; return -6; }
    } else {
        for (int tries = 0; tries < 900 && fd < 0; ++tries) {             // wait for rank 0 (up to 90 s)
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0) {
                off_t len = lseek(fd, 0, SEEK_END);
                if (len < (off_t)s->bytes) { close(fd); fd = -1; }
            }
            if (fd < 0) usleep(100000);
        }
        if (fd < 0) { c->err = "shm communicator: segment of rank 0 not found"; delete s; return -6; }
    }
    s->base = (char*)mmap(nullptr, s->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (s->base == MAP_FAILED) { c->err = "shm communicator: mmap failed"; s->base = nullptr; delete s; return -6; }
    ShmHeader* h = s->hdr();
    if (rank == 0) {                                                         // a fresh segment is zero-filled
        h->red_cap = s->red_cap; h->out_cap = s->out_cap;
        h->nranks = (uint32_t)nranks;                                        // published last: the others wait for it
    } else {
        for (int tries = 0; tries < 900 && ((volatile ShmHeader*)h)->nranks != (uint32_t)nranks; ++tries) usleep(100000);
        if (h->nranks != (uint32_t)nranks || h->red_cap != s->red_cap || h->out_cap != s->out_cap) {
            c->err = "shm communicator: ranks disagree about the segment layout"; munmap(s->base, s->bytes); delete s; return -6;
        }
    }
    for (int k = 0; k < 2 * SHM_MAX_PEERS; ++k) s->dir(rank)[k] = ShmDirEntry{-1, 0, 0};
    c->shm = s;
    c->rank = rank; c->nranks = nranks;
    c->dist = true;
    if (!(getenv("KNP_HALO_OVERLAP") && atoi(getenv("KNP_HALO_OVERLAP")) == 0)) {     // the overlapped apply, as with the halo communicator
        HIPCHK(c, create_halo_stream(c));
        HIPCHK(c, hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->halo_done, hipEventDisableTiming));
    }
    int rc = shm_barrier(c, s);
    if (rc == 0 && rank == 0) shm_unlink(name);                              // everyone is attached: the name can go
    return rc;
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
    if (c->shm) {                                                            // where each peer finds its message in this rank's outbox
        ShmComm* s = (ShmComm*)c->shm;
        if (npeers > SHM_MAX_PEERS) { c->err = "shm communicator: too many peers"; return -1; }
        for (int p = 0; p < npeers; ++p) s->dir(s->rank)[p] = ShmDirEntry{peers[p], c->halo_send_off[p], c->halo_send_cnt[p]};
        return shm_barrier(c, s);
    }
    return 0;
}

// Interface tables of the row-distributed conforming level (see interface_accumulate): peers[p] shares counts[p] conforming dofs with this
// rank; idx = their LOCAL numbers (n_local = size of this rank's level 0), grouped by peer, within a peer in ascending GLOBAL order (the
// peer lists the same dofs in the same order).  uvtx [nuniq]: the distinct shared dofs; aptr / asrc: per distinct dof the positions (into
// idx) of its other owners' values in ascending rank order, with -1 where this rank's own value belongs.  Every rank of the communicator
// calls this (also with npeers = 0).
int knp_amg_interface(knp_ctx* c, int64_t n_local, int npeers, const int32_t* peers, const int64_t* counts, const int32_t* idx,
                      int64_t nuniq, const int32_t* uvtx, const int32_t* aptr, const int32_t* asrc) {
    if (!c || npeers < 0 || nuniq < 0 || n_local < 0) return -1;
    if (!c->dist) { c->err = "amg_interface: no communicator"; return -1; }
    c->if_peer.clear(); c->if_off.clear(); c->if_cnt.clear();
    int64_t total = 0;
    for (int p = 0; p < npeers; ++p) {
        if (peers[p] < 0 || peers[p] >= c->nranks || peers[p] == c->rank || counts[p] <= 0) { c->err = "amg_interface: bad peer entry"; return -1; }
        c->if_peer.push_back(peers[p]);
        c->if_off.push_back(total);
        c->if_cnt.push_back(counts[p]);
        total += counts[p];
    }
    for (int64_t i = 0; i < total; ++i)
        if (idx[i] < 0 || idx[i] >= n_local) { c->err = "amg_interface: dof outside the local level"; return -1; }
    for (int64_t u = 0; u < nuniq; ++u) {
        if (uvtx[u] < 0 || uvtx[u] >= n_local || aptr[u + 1] < aptr[u]) { c->err = "amg_interface: bad shared-dof table"; return -1; }
        int own = 0;
        for (int k = aptr[u]; k < aptr[u + 1]; ++k) {
            if (asrc[k] < -1 || asrc[k] >= total) { c->err = "amg_interface: source position out of range"; return -1; }
            own += asrc[k] < 0;
        }
        if (own != 1) { c->err = "amg_interface: every shared dof needs this rank's own value exactly once"; return -1; }
    }
    if (nuniq && aptr[0] != 0) { c->err = "amg_interface: bad shared-dof table"; return -1; }
    c->if_total = total;
    c->if_nuniq = nuniq;
    hipFree(c->if_idx); hipFree(c->if_uvtx); hipFree(c->if_aptr); hipFree(c->if_asrc); hipFree(c->if_send); hipFree(c->if_recv);
    c->if_idx = c->if_uvtx = c->if_aptr = c->if_asrc = nullptr;
    c->if_send = c->if_recv = nullptr;
    if (total) {
        HIPCHK(c, hipMalloc((void**)&c->if_idx, sizeof(int32_t) * total));
        HIPCHK(c, hipMemcpy(c->if_idx, idx, sizeof(int32_t) * total, hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->if_send, sizeof(double) * total * KNP_MAX_SYS));
        HIPCHK(c, hipMalloc((void**)&c->if_recv, sizeof(double) * total * KNP_MAX_SYS));
    }
    if (nuniq) {
        const int64_t nacc = aptr[nuniq];
        HIPCHK(c, hipMalloc((void**)&c->if_uvtx, sizeof(int32_t) * nuniq));
        HIPCHK(c, hipMemcpy(c->if_uvtx, uvtx, sizeof(int32_t) * nuniq, hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->if_aptr, sizeof(int32_t) * (nuniq + 1)));
        HIPCHK(c, hipMemcpy(c->if_aptr, aptr, sizeof(int32_t) * (nuniq + 1), hipMemcpyHostToDevice));
        HIPCHK(c, hipMalloc((void**)&c->if_asrc, sizeof(int32_t) * (nacc ? nacc : 1)));
        HIPCHK(c, hipMemcpy(c->if_asrc, asrc, sizeof(int32_t) * nacc, hipMemcpyHostToDevice));
    }
    if (c->shm) {                                                            // where each peer finds its message in this rank's outbox
        ShmComm* s = (ShmComm*)c->shm;
        if (npeers > SHM_MAX_PEERS) { c->err = "shm communicator: too many peers"; return -1; }
        for (int k = 0; k < SHM_MAX_PEERS; ++k) s->dir_if(s->rank)[k] = ShmDirEntry{-1, 0, 0};
        for (int p = 0; p < npeers; ++p) s->dir_if(s->rank)[p] = ShmDirEntry{peers[p], c->if_off[p], c->if_cnt[p]};
        return shm_barrier(c, s);
    }
    return 0;
}

}  // extern "C"
