// Multi-GPU layer: communicators (RCCL over xGMI, or a host-staged callback transport), the
// row-block partition plan, and the distributed Sparse operator.
//
// The reference is a single-process CPU code: everything here is new design for
// one process per MI355X.  Data path collectives per GCR iteration:
//   * 1 halo exchange per SpMV: ncclSend/ncclRecv pairs inside one group on a dedicated stream,
//     overlapped with the rows that touch no remote column;
//   * 2 all-reduces of a handful of doubles (4, then 1 + 2*lim), in place on device memory, on the
//     compute stream — results never visit the host.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <map>

#include "internal.h"
#include "reduce.h"
#include "pw_tail_dev.h"

namespace mgcr {

// ------------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ------------------------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi &rccl() {
    static RcclApi api;
    return api;
}

static int rccl_load() {
    RcclApi &a = rccl();
    if (a.handle) return MGCR_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
    }
    MGCR_CHECK(a.handle, MGCR_ERR_COMM, "cannot dlopen librccl.so.1: %s", dlerror());
#define SYM(field, name)                                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name));       \
    MGCR_CHECK(a.field, MGCR_ERR_COMM, "librccl lacks symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    return MGCR_OK;
}

#define MGCR_NCCL(call)                                                                              \
    do {                                                                                             \
        ncclResult_t r__ = (call);                                                                   \
        if (r__ != ncclSuccess) {                                                                    \
            set_error("RCCL error %d (%s) in %s", (int)r__, rccl().GetErrorString(r__), #call);       \
            return MGCR_ERR_COMM;                                                                    \
        }                                                                                            \
    } while (0)

// ------------------------------------------------------------------------------------------------
struct Comm {
    int rank = 0, nranks = 1;
    bool is_rccl = false;
    ncclComm_t nccl = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    mgcr_allreduce_cb allreduce = nullptr;
    mgcr_exchange_cb exchange = nullptr;
    void *user = nullptr;
    // staging for host-level collectives over RCCL (set-up only)
    double *d_stage = nullptr;
    size_t d_stage_cap = 0;
    double *h_pin = nullptr;  // pinned, for the host-staged transport's scalar all-reduces
    // peer-write all-reduce of the per-iteration scalars (below): mailboxes mapped into every rank
    bool pw_tried = false, pw_on = false;
    uint64_t *pw_mbox = nullptr;             // this rank's mailbox (uncached device memory, shared by hipIpc)
    uint64_t *pw_peer[PW_MAX_RANKS] = {};    // pw_peer[r]: rank r's mailbox as mapped here (own one for r == rank)
    uint32_t pw_seq = 0;                     // sequence number of the last all-reduce (never 0 on the wire)
    int *pw_err = nullptr;                   // pinned host word the kernel sets when a wait timed out
    unsigned *pw_ticket = nullptr;           // device counter of the producer kernels' fold tails (pw_tail_dev.h)
};

// live communicators: every host synchronisation point that hands distributed results back asks each of them whether a
// peer-write wait timed out since the last look (comm_check_all) — an operator apply, a V-cycle or a nested solve on a
// distributed operator must not return plausible numbers computed from a halo that never arrived
static std::vector<Comm *> &live_comms() {
    static std::vector<Comm *> v;
    return v;
}
static std::mutex &live_comms_mtx() {
    static std::mutex m;
    return m;
}
static void comm_register(Comm *c) {
    std::lock_guard<std::mutex> lk(live_comms_mtx());
    live_comms().push_back(c);
}
static void comm_unregister(Comm *c) {
    std::lock_guard<std::mutex> lk(live_comms_mtx());
    auto &v = live_comms();
    v.erase(std::remove(v.begin(), v.end(), c), v.end());
}

// first sequence number of the peer-write exchanges (tests start just below the 32-bit wrap: MGCR_TEST_PW_SEQ0)
static uint32_t pw_seq0() {
    const char *e = getenv("MGCR_TEST_PW_SEQ0");
    return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u;
}
// Next sequence number.  Never 0 (mailboxes start zeroed), and the slot parity (seq & 1) must keep alternating: after
// 0xFFFFFFFF (odd) comes 2, not 1 — two consecutive exchanges in one slot would let a fast rank overwrite words a slower
// peer has not read yet.
static uint32_t pw_advance(uint32_t seq) {
    seq++;
    return seq == 0 ? 2u : seq;
}

static int comm_device_ready(Comm *c) {
    if (c->comm_stream) return MGCR_OK;
    MGCR_TRY(require_ctx());
    MGCR_HIP(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    MGCR_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    MGCR_HIP(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    MGCR_HIP(hipHostMalloc((void **)&c->h_pin, sizeof(double) * 1024, hipHostMallocDefault));
    return MGCR_OK;
}

static int stage_reserve(Comm *c, size_t doubles) {
    if (doubles <= c->d_stage_cap) return MGCR_OK;
    if (c->d_stage) hipFree(c->d_stage);
    c->d_stage = nullptr;
    MGCR_HIP(hipMalloc((void **)&c->d_stage, sizeof(double) * doubles));
    c->d_stage_cap = doubles;
    return MGCR_OK;
}


// ------------------------------------------------------------------------------------------------
// Peer-write all-reduce for the scalars of a GCR step (4, then 1 + 2*lim doubles).
//
// An RCCL all-reduce of a few doubles costs its launch plus a ring/tree protocol, twice per iteration on
// the critical path between dependent kernels.  Here the kernel that folds the workgroup partials (one
// wave per scalar, blas1.hip fold_kernel) also exchanges them: lane j stores the wave's sum straight into
// rank j's mailbox over xGMI and polls its own mailbox for rank j's sum; the wave then adds the values in
// RANK ORDER, so every rank obtains the same bits.  No further launch, no second pass.
//
// Wire format (the "LL" idea: data and flag travel in ONE 8-byte store, which the fabric never tears, so
// no ordering between separate stores is needed): a double goes as two words {lo32 | seq<<32},
// {hi32 | seq<<32}; the receiver spins until both words carry the expected sequence number.  Mailboxes
// are zero-initialised and seq is never 0.  Two slots (seq & 1): a rank finishes all-reduce s only after
// every peer has SENT s, i.e. after every peer finished s-1, so nobody can still be reading the slot that
// s+1 overwrites.  Mailboxes are uncached (fine-grained) device memory shared with hipIpc handles — the
// allocation class RCCL itself uses for its peer buffers.
//
// The path validates itself when the communicator's first distributed operator is created (a few
// all-reduces with known answers, every wait bounded by a wall-clock limit); if any rank fails to map a
// mailbox, times out or sees a wrong sum, ALL ranks fall back to RCCL (or the host transport).
// MGCR_PEER_ALLREDUCE=0 turns it off.
// ------------------------------------------------------------------------------------------------
struct PwPeers {
    uint64_t *mb[PW_MAX_RANKS];
};
constexpr size_t PW_MBOX_WORDS = (size_t)2 * PW_MAX_RANKS * PW_MAX_SCALARS * 2;
// wall_clock64 runs at 100 MHz.  Self-tests (ranks just synchronised by a set-up collective): 3 s.  Production: 20 s — the
// ranks of one solve may arrive skewed (one of them still reading a file), but a wave must never spin anywhere near the
// driver's compute-queue watchdog (60 s).
constexpr long long PW_TIMEOUT_TEST = 300000000LL;
static long long pw_timeout_run() {  // MGCR_PEER_TIMEOUT_MS (tests shorten it), clamped to 1 ms .. 30 s
    static const long long ticks = [] {
        long long ms = 20000;
        if (const char *e = getenv("MGCR_PEER_TIMEOUT_MS")) ms = atoll(e);
        ms = ms < 1 ? 1 : ms > 30000 ? 30000 : ms;
        return ms * 100000LL;
    }();
    return ticks;
}

__global__ void __launch_bounds__(64) fold_pw_kernel(const double *__restrict__ pa, int na, const double *__restrict__ pb, int nb,
                                                     double *__restrict__ out, int nblk, PwPeers peers, int rank, int nranks,
                                                     uint32_t seq, int *err, long long timeout) {
    const int k = blockIdx.x, lane = threadIdx.x;
    double acc;
    if (nblk > 0) {  // fold scalar k exactly as fold_kernel does
        const double *src = k < na ? pa + (size_t)k * RED_MAX_BLOCKS : pb + (size_t)(k - na) * RED_MAX_BLOCKS;
        acc = wave_fold_slab(src, nblk);
    } else {
        acc = out[k];  // already folded: all-reduce in place
    }
    const double tot = pw_exchange_scalar(peers.mb, rank, nranks, seq, err, timeout, k, acc);   // (pw_tail_dev.h: shared with the producer kernels' tails)
    if (lane == 0) out[k] = tot;
}

static uint32_t pw_next_seq(Comm *c) {
    c->pw_seq = pw_advance(c->pw_seq);
    return c->pw_seq;
}

static int pw_launch(Comm *c, const double *pa, int na, const double *pb, int nb, double *out, int nblk) {
    PwPeers peers;
    for (int r = 0; r < PW_MAX_RANKS; r++) peers.mb[r] = c->pw_peer[r < c->nranks ? r : c->rank];
    hipLaunchKernelGGL(fold_pw_kernel, dim3(na + nb), dim3(64), 0, ctx().stream, pa, na, pb, nb, out, nblk, peers, c->rank, c->nranks,
                       pw_next_seq(c), c->pw_err, c->pw_on ? pw_timeout_run() : PW_TIMEOUT_TEST);
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}

static int comm_allreduce_host(Comm *c, double *buf, int64_t count);

// The fold + exchange of a reduction inside the kernel that produces its partials (pw_tail_dev.h)
static int g_pw_tail = -1;
static int64_t g_pw_tail_count = 0;
static bool pw_tail_enabled() {
    if (g_pw_tail < 0) g_pw_tail = !(getenv("MGCR_PW_TAIL") && atoi(getenv("MGCR_PW_TAIL")) == 0);
    return g_pw_tail != 0;
}
bool set_pw_tail_enabled(bool on) {
    const bool prev = pw_tail_enabled();
    g_pw_tail = on ? 1 : 0;
    return prev;
}
int64_t comm_pw_tail_count() { return g_pw_tail_count; }
bool comm_pw_tail_begin(Comm *c, PwTail *t) {
    if (!c || !c->pw_on || !pw_tail_enabled() || c->nranks < 2) return false;
    if (!c->pw_ticket) {
        if (hipMalloc((void **)&c->pw_ticket, sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); c->pw_ticket = nullptr; return false; }
        if (hipMemsetAsync(c->pw_ticket, 0, sizeof(unsigned), ctx().stream) != hipSuccess) return false;
    }
    for (int r = 0; r < PW_MAX_RANKS; r++) t->mb[r] = c->pw_peer[r < c->nranks ? r : c->rank];
    t->rank = c->rank;
    t->nranks = c->nranks;
    t->seq = pw_next_seq(c);
    t->err = c->pw_err;
    t->timeout = pw_timeout_run();
    t->ticket = c->pw_ticket;
    g_pw_tail_count++;
    return true;
}

static void pw_release(Comm *c) {
    for (int r = 0; r < c->nranks && r < PW_MAX_RANKS; r++)
        if (r != c->rank && c->pw_peer[r]) hipIpcCloseMemHandle(c->pw_peer[r]);
    for (int r = 0; r < PW_MAX_RANKS; r++) c->pw_peer[r] = nullptr;
    if (c->pw_mbox) hipFree(c->pw_mbox);
    c->pw_mbox = nullptr;
    if (c->pw_err) hipHostFree(c->pw_err);
    c->pw_err = nullptr;
    if (c->pw_ticket) hipFree(c->pw_ticket);
    c->pw_ticket = nullptr;
    c->pw_on = false;
}

// Collective (every rank of the communicator calls it at the same point): map the mailboxes, run the
// self-test, agree on the outcome.  Never fails the caller: on any problem the communicator simply keeps
// its RCCL / host all-reduce.
static int comm_pw_setup(Comm *c) {
    if (c->pw_tried) return MGCR_OK;
    c->pw_tried = true;
    if (c->nranks < 2 || c->nranks > PW_MAX_RANKS) return MGCR_OK;
    if (getenv("MGCR_PEER_ALLREDUCE") && atoi(getenv("MGCR_PEER_ALLREDUCE")) == 0) return MGCR_OK;
    MGCR_TRY(comm_device_ready(c));
    const int nr = c->nranks;
    bool ok = true;
    hipIpcMemHandle_t mine;
    memset(&mine, 0, sizeof(mine));
    // own mailbox: uncached device memory, zeroed BEFORE anybody can learn its handle
    if (hipExtMallocWithFlags((void **)&c->pw_mbox, PW_MBOX_WORDS * sizeof(uint64_t), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        c->pw_mbox = nullptr;
        ok = false;
    }
    if (ok && hipHostMalloc((void **)&c->pw_err, sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); c->pw_err = nullptr; ok = false; }
    if (ok) {
        *c->pw_err = 0;
        ok = hipMemset(c->pw_mbox, 0, PW_MBOX_WORDS * sizeof(uint64_t)) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
             hipIpcGetMemHandle(&mine, c->pw_mbox) == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    // all-gather of the handles (one double per byte: the set-up all-reduce sums doubles) + "I am fine" count
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t size");
    std::vector<double> g((size_t)nr * 64 + 1, 0.);
    if (ok) {
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&mine);
        for (int i = 0; i < 64; i++) g[(size_t)c->rank * 64 + i] = (double)b[i];
        g[(size_t)nr * 64] = 1.;
    }
    MGCR_TRY(comm_allreduce_host(c, g.data(), (int64_t)g.size()));
    bool all = (int)g[(size_t)nr * 64] == nr;
    bool mapped = all;
    if (all) {
        for (int r = 0; r < nr && mapped; r++) {
            if (r == c->rank) { c->pw_peer[r] = c->pw_mbox; continue; }
            hipIpcMemHandle_t h;
            unsigned char *b = reinterpret_cast<unsigned char *>(&h);
            for (int i = 0; i < 64; i++) b[i] = (unsigned char)g[(size_t)r * 64 + i];
            void *ptr = nullptr;
            if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); mapped = false; break; }
            c->pw_peer[r] = (uint64_t *)ptr;
        }
    }
    double flag = mapped ? 0. : 1.;
    MGCR_TRY(comm_allreduce_host(c, &flag, 1));   // nobody writes into a mailbox before everybody has mapped all of them
    bool good = all && flag == 0.;
    if (good) {
        // self-test: PW_TEST rounds over PW_TEST_N scalars with known sums, through the production kernel
        constexpr int PW_TEST = 6, PW_TEST_N = 33;
        double *d = nullptr;
        good = hipMalloc((void **)&d, sizeof(double) * PW_TEST_N) == hipSuccess;
        std::vector<double> h((size_t)PW_TEST_N);
        for (int t = 0; t < PW_TEST && good; t++) {
            for (int k = 0; k < PW_TEST_N; k++) h[(size_t)k] = 1.0 / (double)(1 + c->rank + 3 * k + 7 * t);
            good = hipMemcpy(d, h.data(), sizeof(double) * PW_TEST_N, hipMemcpyHostToDevice) == hipSuccess;
            if (good) good = pw_launch(c, nullptr, PW_TEST_N, nullptr, 0, d, 0) == MGCR_OK;
            if (good) good = hipStreamSynchronize(ctx().stream) == hipSuccess && hipMemcpy(h.data(), d, sizeof(double) * PW_TEST_N, hipMemcpyDeviceToHost) == hipSuccess;
            for (int k = 0; k < PW_TEST_N && good; k++) {
                double want = 0.;
                for (int r = 0; r < nr; r++) want += 1.0 / (double)(1 + r + 3 * k + 7 * t);
                good = h[(size_t)k] == want;
            }
            if (good) good = *(volatile int *)c->pw_err == 0;
        }
        if (!good) (void)hipGetLastError();
        if (d) hipFree(d);
    }
    flag = good ? 0. : 1.;
    MGCR_TRY(comm_allreduce_host(c, &flag, 1));
    if (flag == 0.) {
        c->pw_on = true;
    } else {
        hipDeviceSynchronize();
        pw_release(c);
        (void)hipGetLastError();
    }
    return MGCR_OK;
}

bool comm_collectives(Comm *c);

// host-level all-reduce (set-up): in-place sum of `count` doubles over all ranks
static int comm_allreduce_host(Comm *c, double *buf, int64_t count) {
    if (c->nranks == 1 && !(c->is_rccl && comm_collectives(c))) return MGCR_OK;
    if (!c->is_rccl) {
        int rc = c->allreduce(c->user, buf, count);
        MGCR_CHECK(rc == 0, MGCR_ERR_COMM, "allreduce callback failed (%d)", rc);
        return MGCR_OK;
    }
    MGCR_TRY(comm_device_ready(c));
    MGCR_TRY(stage_reserve(c, (size_t)count));
    hipStream_t st = ctx().stream;
    MGCR_HIP(hipMemcpyAsync(c->d_stage, buf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st));
    MGCR_NCCL(rccl().AllReduce(c->d_stage, c->d_stage, (size_t)count, ncclDouble, ncclSum, c->nccl, st));
    MGCR_HIP(hipMemcpyAsync(buf, c->d_stage, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st));
    MGCR_HIP(hipStreamSynchronize(st));
    return MGCR_OK;
}

// host-level neighbour exchange (set-up): counts in doubles
static int comm_exchange_host(Comm *c, int npeers, const int *peers, const double *const *send, const int64_t *scount,
                              double *const *recv, const int64_t *rcount) {
    if (npeers == 0) return MGCR_OK;
    if (!c->is_rccl) {
        int rc = c->exchange(c->user, npeers, peers, send, scount, recv, rcount);
        MGCR_CHECK(rc == 0, MGCR_ERR_COMM, "exchange callback failed (%d)", rc);
        return MGCR_OK;
    }
    MGCR_TRY(comm_device_ready(c));
    size_t tot = 0;
    for (int p = 0; p < npeers; p++) tot += (size_t)scount[p] + (size_t)rcount[p];
    MGCR_TRY(stage_reserve(c, tot));
    hipStream_t st = ctx().stream;
    std::vector<double *> ds((size_t)npeers), dr((size_t)npeers);
    size_t off = 0;
    for (int p = 0; p < npeers; p++) {
        ds[(size_t)p] = c->d_stage + off; off += (size_t)scount[p];
        dr[(size_t)p] = c->d_stage + off; off += (size_t)rcount[p];
        if (scount[p]) MGCR_HIP(hipMemcpyAsync(ds[(size_t)p], send[p], sizeof(double) * (size_t)scount[p], hipMemcpyHostToDevice, st));
    }
    MGCR_NCCL(rccl().GroupStart());
    for (int p = 0; p < npeers; p++) {
        if (scount[p]) MGCR_NCCL(rccl().Send(ds[(size_t)p], (size_t)scount[p], ncclDouble, peers[p], c->nccl, st));
        if (rcount[p]) MGCR_NCCL(rccl().Recv(dr[(size_t)p], (size_t)rcount[p], ncclDouble, peers[p], c->nccl, st));
    }
    MGCR_NCCL(rccl().GroupEnd());
    for (int p = 0; p < npeers; p++)
        if (rcount[p]) MGCR_HIP(hipMemcpyAsync(recv[p], dr[(size_t)p], sizeof(double) * (size_t)rcount[p], hipMemcpyDeviceToHost, st));
    MGCR_HIP(hipStreamSynchronize(st));
    return MGCR_OK;
}

// ------------------------------------------------------------------------------------------------
// partition plan (host only)
// ------------------------------------------------------------------------------------------------
struct Plan {
    Comm *comm = nullptr;
    int64_t n_global = 0, row0 = 0, nloc = 0, nnz = 0;
    std::vector<int64_t> offsets;       // row0 of every rank, + n_global
    std::vector<int64_t> halo_gid;      // sorted global ids of the remote columns (grouped by owner, ascending)
    std::vector<int32_t> peers;         // ranks exchanged with (ascending)
    std::vector<int64_t> recv_count, recv_off;            // per peer: entries of the halo segment
    std::vector<std::vector<int64_t>> send_rows;          // per peer: my local rows it needs, in its halo order
    std::vector<int64_t> col_local;     // nnz
    int64_t interior_begin = 0, interior_end = 0;         // rows [begin, end) reference no halo column
};

bool comm_collectives(Comm *c);

static int owner_of(const std::vector<int64_t> &offsets, int64_t gid) {
    return (int)(std::upper_bound(offsets.begin(), offsets.end(), gid) - offsets.begin()) - 1;
}

static int plan_build(Comm *c, int64_t n_global, int64_t row0, int64_t nloc, const int64_t *rowptr, const int64_t *col, Plan **out) {
    MGCR_CHECK(c && rowptr && row0 >= 0 && nloc >= 0 && row0 + nloc <= n_global, MGCR_ERR_INVALID, "mgcr_plan_create: bad row block");
    Plan *P = new Plan();
    P->comm = c; P->n_global = n_global; P->row0 = row0; P->nloc = nloc; P->nnz = rowptr[nloc];
    const int R = c->nranks;
    // row offsets of all ranks
    std::vector<double> tmp((size_t)R + 0, 0.);
    tmp[(size_t)c->rank] = (double)row0;
    int rc = comm_allreduce_host(c, tmp.data(), R);
    if (rc != MGCR_OK) { delete P; return rc; }
    P->offsets.resize((size_t)R + 1);
    for (int r = 0; r < R; r++) P->offsets[(size_t)r] = (int64_t)tmp[(size_t)r];
    P->offsets[(size_t)R] = n_global;
    for (int r = 0; r < R; r++)
        if (P->offsets[(size_t)r] > P->offsets[(size_t)r + 1]) {
            delete P;
            set_error("mgcr_plan_create: row blocks must be ordered by rank and contiguous");
            return MGCR_ERR_INVALID;
        }
    // remote columns
    std::vector<int64_t> remote;
    for (int64_t l = 0; l < P->nnz; l++) {
        int64_t g = col[l];
        if (g < 0 || g >= n_global) { delete P; set_error("mgcr_plan_create: column %lld out of range", (long long)g); return MGCR_ERR_INVALID; }
        if (g < row0 || g >= row0 + nloc) remote.push_back(g);
    }
    std::sort(remote.begin(), remote.end());
    remote.erase(std::unique(remote.begin(), remote.end()), remote.end());
    P->halo_gid = remote;  // ascending global id == grouped by owner rank
    std::map<int64_t, int64_t> slot;
    for (size_t h = 0; h < remote.size(); h++) slot[remote[h]] = (int64_t)h;
    // what I need from whom
    std::vector<int64_t> need_cnt((size_t)R, 0);
    for (int64_t g : remote) need_cnt[(size_t)owner_of(P->offsets, g)]++;
    // counts matrix: cnt[r][q] = number of entries rank r needs from rank q
    std::vector<double> M((size_t)R * R, 0.);
    for (int q = 0; q < R; q++) M[(size_t)c->rank * R + q] = (double)need_cnt[(size_t)q];
    rc = comm_allreduce_host(c, M.data(), (int64_t)R * R);
    if (rc != MGCR_OK) { delete P; return rc; }
    for (int q = 0; q < R; q++) {
        if (q == c->rank) continue;
        int64_t rc_ = (int64_t)M[(size_t)c->rank * R + q], sc_ = (int64_t)M[(size_t)q * R + c->rank];
        if (rc_ || sc_) P->peers.push_back(q);
    }
    const int np = (int)P->peers.size();
    P->recv_count.assign((size_t)np, 0);
    P->recv_off.assign((size_t)np, 0);
    P->send_rows.assign((size_t)np, {});
    std::vector<int64_t> send_count((size_t)np, 0);
    int64_t off = 0;
    for (int p = 0; p < np; p++) {
        int q = P->peers[(size_t)p];
        P->recv_count[(size_t)p] = (int64_t)M[(size_t)c->rank * R + q];
        P->recv_off[(size_t)p] = off;
        off += P->recv_count[(size_t)p];
        send_count[(size_t)p] = (int64_t)M[(size_t)q * R + c->rank];
        P->send_rows[(size_t)p].resize((size_t)send_count[(size_t)p]);
    }
    // tell every peer which of its rows I need (global ids, sent as bit patterns in doubles)
    {
        std::vector<const double *> sp((size_t)np);
        std::vector<double *> rp((size_t)np);
        std::vector<int64_t> sc((size_t)np), rcv((size_t)np);
        for (int p = 0; p < np; p++) {
            sp[(size_t)p] = reinterpret_cast<const double *>(P->halo_gid.data() + P->recv_off[(size_t)p]);
            sc[(size_t)p] = P->recv_count[(size_t)p];
            rp[(size_t)p] = reinterpret_cast<double *>(P->send_rows[(size_t)p].data());
            rcv[(size_t)p] = send_count[(size_t)p];
        }
        rc = comm_exchange_host(c, np, P->peers.data(), sp.data(), sc.data(), rp.data(), rcv.data());
        if (rc != MGCR_OK) { delete P; return rc; }
        for (int p = 0; p < np; p++)
            for (int64_t &g : P->send_rows[(size_t)p]) {
                if (g < row0 || g >= row0 + nloc) { delete P; set_error("mgcr_plan_create: peer asked for a row this rank does not own"); return MGCR_ERR_COMM; }
                g -= row0;
            }
    }
    // local column numbering and the interior row range
    P->col_local.resize((size_t)P->nnz);
    std::vector<char> touches((size_t)nloc, 0);
    for (int64_t r = 0; r < nloc; r++)
        for (int64_t l = rowptr[r]; l < rowptr[r + 1]; l++) {
            int64_t g = col[l];
            if (g >= row0 && g < row0 + nloc) P->col_local[(size_t)l] = g - row0;
            else { P->col_local[(size_t)l] = nloc + slot[g]; touches[(size_t)r] = 1; }
        }
    // longest run of rows without halo columns (for a slab partition: everything but the first and last plane)
    int64_t best_b = 0, best_e = 0, cur_b = 0;
    for (int64_t r = 0; r <= nloc; r++) {
        if (r == nloc || touches[(size_t)r]) {
            if (r - cur_b > best_e - best_b) { best_b = cur_b; best_e = r; }
            cur_b = r + 1;
        }
    }
    P->interior_begin = best_b;
    P->interior_end = best_e;
    *out = P;
    return MGCR_OK;
}

// ------------------------------------------------------------------------------------------------
// distributed operator state
// ------------------------------------------------------------------------------------------------
struct DistCsr {
    Comm *comm = nullptr;
    Plan *plan = nullptr;
    cplx *xh = nullptr;        // halo segment [n_halo]
    cplx *sendbuf = nullptr;   // packed send data (all peers)
    int32_t *send_idx = nullptr;
    std::vector<int64_t> send_off, send_cnt;
    std::vector<int64_t> send_contig;  // >= 0: the peer's rows are the contiguous range starting here (no packing)
    std::vector<double> h_send, h_recv;  // host staging (callback transport)
    // peer-write halo exchange (below): receive slots mapped into the neighbours
    bool pw_on = false;
    unsigned char *pw_rx = nullptr;        // own: [2 slots][n_halo] cplx, then [2][PW_MAX_RANKS] flag words (uncached, hipIpc)
    std::vector<unsigned char *> pw_peer_rx;  // per peer: its pw_rx as mapped here
    struct HaloPwPeer *pw_tab = nullptr;   // device table, one entry per peer
    int *pw_ticket = nullptr;              // device: workgroups of the running exchange that have stored their rows
    uint32_t pw_seq = 0;
    unsigned pw_grid_x = 1;
    bool pw_wait_pending = false;          // a split exchange has stored and published; its wait kernel is still to be launched (dist_halo_end)
};

static bool halo_overlap() {
    static const bool on = getenv("MGCR_HALO_OVERLAP") && atoi(getenv("MGCR_HALO_OVERLAP")) != 0;
    return on;
}
bool dist_halo_overlaps() { return halo_overlap(); }


__global__ void __launch_bounds__(256) pack_kernel(int64_t n, const int32_t *__restrict__ idx, const cplx *__restrict__ x,
                                                   cplx *__restrict__ out, const int *__restrict__ skip, int skip_it) {
    if (skip && skip[0] < skip[1] + skip_it) return;  // {stop_at, base}: see gcr.hip DevState
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = x[idx[i]];
}


// ------------------------------------------------------------------------------------------------
// Peer-write halo exchange.  With RCCL one exchange is an ncclSend/ncclRecv group: a kernel launch plus
// its handshake on the critical path of every operator apply, for a payload of one grid plane
// (262 KB at 128^3).  Here ONE kernel does it: its workgroups store this rank's boundary rows straight
// into the neighbours' receive slots over xGMI (plain 16-byte stores into uncached, hipIpc-mapped memory),
// fence, and take a ticket; the last workgroup then publishes the sequence number in every neighbour's
// flag word (release, system scope) and waits — bounded — for the neighbours' flags in its own.  When the
// kernel retires the halo has arrived, and the apply kernel reads it in place from the receive slot.
// Two slots (seq & 1): a neighbour publishes s+1 only after its apply s has run (stream order), and this
// rank starts s+2 only after it has seen the neighbour's s+1, so the slot that s+2 overwrites is free.
// The peer lists are symmetric (A lists B iff B lists A: one's send is the other's receive).
// Validated by a self-test at dist_csr_create (global row numbers through both slots); on any failure
// all ranks keep the RCCL / host exchange.  MGCR_PEER_HALO=0 turns it off.
// ------------------------------------------------------------------------------------------------
struct HaloPwPeer {
    cplx *dst[2];                 // where this rank's rows land in the peer's receive slots
    uint64_t *flag_remote[2];     // the peer's flag word for this rank
    const uint64_t *flag_local[2];  // this rank's flag word for the peer
    int64_t send_off, send_cnt;   // this rank's send list for the peer (send_idx)
    cplx *rx_local[2];            // where the peer's rows land in this rank's receive slots ...
    int64_t recv_cnt;             // ... and how many: poisoned with NaN when the peer never arrives
};

// what the last workgroup of an exchange (or the wait kernel) does for peer q: wait — bounded — for the neighbour's flag
__device__ __forceinline__ void halo_pw_wait_peer(const HaloPwPeer &q, int slot, uint32_t seq, int *err, long long timeout) {
    const long long t0 = wall_clock64();
    bool ok = false;
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        for (;;) {
            if (__hip_atomic_load(q.flag_local[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == (uint64_t)seq) { ok = true; break; }
            if (wall_clock64() - t0 > timeout) break;
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (!ok) {
        // the neighbour never published: flag it (every host synchronisation point turns the flag into MGCR_ERR_COMM,
        // comm_check_all) and poison the rows it owed with NaN, as the all-reduce does with its sums — a missed
        // check must not be able to yield plausible numbers from a stale or half-written slot
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        for (int64_t i = 0; i < q.recv_cnt; i++) q.rx_local[slot][i] = make_double2(nan, nan);
    }
}

// WAIT = false: the exchange is SPLIT — this kernel stores and publishes, and halo_pw_wait_kernel, launched after the rows
// that need no halo have been multiplied, waits for the neighbours (spmv.hip csr_apply_t): the wait — the neighbour's own
// kernels plus the link — then overlaps with the interior rows instead of preceding them.  Same protocol: the wait kernel is
// never skipped and precedes this rank's next exchange in stream order, so every exchange remains a rendezvous.
template <bool WAIT>
__global__ void __launch_bounds__(256) halo_pw_kernel(const HaloPwPeer *__restrict__ tab, int npeer, const int32_t *__restrict__ idx,
                                                      const cplx *__restrict__ x, uint32_t seq, int *ticket, int *err, long long timeout) {
    const int p = blockIdx.y, slot = (int)(seq & 1u);
    const HaloPwPeer pe = tab[p];
    cplx *dst = pe.dst[slot];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pe.send_cnt; i += (int64_t)gridDim.x * 256)
        dst[i] = x[idx[pe.send_off + i]];
    __threadfence_system();   // this thread's remote stores have landed
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)(gridDim.x * gridDim.y) - 1;
    __syncthreads();
    if (!last) return;
    // every workgroup's rows are in place: publish, then wait for the neighbours
    if ((int)threadIdx.x < npeer) {
        const HaloPwPeer q = tab[threadIdx.x];
        __hip_atomic_store(q.flag_remote[slot], (uint64_t)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (WAIT) halo_pw_wait_peer(q, slot, seq, err, timeout);
    }
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// second half of a split exchange: one wave, lane p waits for peer p
__global__ void __launch_bounds__(64) halo_pw_wait_kernel(const HaloPwPeer *__restrict__ tab, int npeer, uint32_t seq, int *err, long long timeout) {
    if ((int)threadIdx.x < npeer) halo_pw_wait_peer(tab[threadIdx.x], (int)(seq & 1u), seq, err, timeout);
}

__global__ void __launch_bounds__(256) halo_test_fill_kernel(cplx *x, int64_t n, int64_t row0, double im) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = make_double2((double)(row0 + i), im);
}

static size_t halo_pw_slot_bytes(const DistCsr *d) { return (d->plan->halo_gid.size() * sizeof(cplx) + 255) / 256 * 256; }

static const cplx *halo_pw_slot(const DistCsr *d, uint32_t seq) {
    return reinterpret_cast<const cplx *>(d->pw_rx + (size_t)(seq & 1u) * halo_pw_slot_bytes(d));
}

static int g_halo_split = -1;
static bool halo_split_enabled() {   // MGCR_HALO_SPLIT=0 / mgcr_set_option("halo_split", 0): the stand-alone apply waits for its halo before any row, as the fused GCR steps do
    if (g_halo_split < 0) g_halo_split = !(getenv("MGCR_HALO_SPLIT") && atoi(getenv("MGCR_HALO_SPLIT")) == 0);
    return g_halo_split != 0;
}
bool set_halo_split(bool on) {
    const bool prev = halo_split_enabled();
    g_halo_split = on ? 1 : 0;
    return prev;
}
static int halo_pw_launch(DistCsr *d, const cplx *x, bool split = false) {
    Comm *c = d->comm;
    d->pw_seq = pw_advance(d->pw_seq);
    const int np = (int)d->plan->peers.size();
    MGCR_CHECK(np <= 64, MGCR_ERR_UNSUPPORTED, "peer-write halo exchange: at most 64 neighbours");
    const long long timeout = d->pw_on ? pw_timeout_run() : PW_TIMEOUT_TEST;
    if (split && halo_split_enabled()) {
        hipLaunchKernelGGL(halo_pw_kernel<false>, dim3(d->pw_grid_x, (unsigned)np), dim3(256), 0, ctx().stream, (const HaloPwPeer *)d->pw_tab, np,
                           (const int32_t *)d->send_idx, x, d->pw_seq, d->pw_ticket, c->pw_err, timeout);
        d->pw_wait_pending = true;
    } else {
        hipLaunchKernelGGL(halo_pw_kernel<true>, dim3(d->pw_grid_x, (unsigned)np), dim3(256), 0, ctx().stream, (const HaloPwPeer *)d->pw_tab, np,
                           (const int32_t *)d->send_idx, x, d->pw_seq, d->pw_ticket, c->pw_err, timeout);
    }
    MGCR_HIP(hipGetLastError());
    return MGCR_OK;
}
static int64_t g_halo_split_count = 0;
int64_t dist_halo_split_count() { return g_halo_split_count; }

static void halo_pw_release(DistCsr *d) {
    for (unsigned char *q : d->pw_peer_rx)
        if (q) hipIpcCloseMemHandle(q);
    d->pw_peer_rx.clear();
    if (d->pw_rx) hipFree(d->pw_rx);
    if (d->pw_tab) hipFree(d->pw_tab);
    if (d->pw_ticket) hipFree(d->pw_ticket);
    d->pw_rx = nullptr; d->pw_tab = nullptr; d->pw_ticket = nullptr;
    d->pw_on = false;
}

// Collective over the communicator, called by dist_csr_create once the send lists are on the device.
// Never fails the caller for a transport reason: on any problem every rank keeps the RCCL / host exchange.
static int halo_pw_setup(DistCsr *d) {
    Comm *c = d->comm;
    Plan *P = d->plan;
    if (!c->pw_on) return MGCR_OK;   // same mechanism as the peer-write all-reduce: only where that one validated
    if (getenv("MGCR_PEER_HALO") && atoi(getenv("MGCR_PEER_HALO")) == 0) return MGCR_OK;
    const int nr = c->nranks, np = (int)P->peers.size();
    const size_t nh = P->halo_gid.size(), slot_bytes = halo_pw_slot_bytes(d);
    const size_t flag_off = 2 * slot_bytes, total = flag_off + 2 * PW_MAX_RANKS * sizeof(uint64_t);
    bool ok = true;
    hipIpcMemHandle_t mine;
    memset(&mine, 0, sizeof(mine));
    if (hipExtMallocWithFlags((void **)&d->pw_rx, total, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); d->pw_rx = nullptr; ok = false; }
    if (ok) ok = hipMemset(d->pw_rx, 0, total) == hipSuccess && hipDeviceSynchronize() == hipSuccess && hipIpcGetMemHandle(&mine, d->pw_rx) == hipSuccess;
    if (ok) ok = hipMalloc((void **)&d->pw_ticket, sizeof(int)) == hipSuccess && hipMemset(d->pw_ticket, 0, sizeof(int)) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    std::vector<double> g((size_t)nr * 64 + 1, 0.);
    if (ok) {
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&mine);
        for (int i = 0; i < 64; i++) g[(size_t)c->rank * 64 + i] = (double)b[i];
        g[(size_t)nr * 64] = 1.;
    }
    MGCR_TRY(comm_allreduce_host(c, g.data(), (int64_t)g.size()));
    bool good = (int)g[(size_t)nr * 64] == nr;
    // where my rows start in each neighbour's halo segment
    std::vector<double> my_off((size_t)np), their_off((size_t)np, 0.);
    {
        std::vector<const double *> sp((size_t)np);
        std::vector<double *> rp((size_t)np);
        std::vector<int64_t> one((size_t)np, 1);
        for (int p = 0; p < np; p++) { my_off[(size_t)p] = (double)P->recv_off[(size_t)p]; sp[(size_t)p] = &my_off[(size_t)p]; rp[(size_t)p] = &their_off[(size_t)p]; }
        MGCR_TRY(comm_exchange_host(c, np, P->peers.data(), sp.data(), one.data(), rp.data(), one.data()));
    }
    bool mapped = good;
    std::vector<HaloPwPeer> tab((size_t)np);
    int64_t max_cnt = 0;
    if (good) {
        d->pw_peer_rx.assign((size_t)np, nullptr);
        for (int p = 0; p < np && mapped; p++) {
            const int r = P->peers[(size_t)p];
            hipIpcMemHandle_t h;
            unsigned char *b = reinterpret_cast<unsigned char *>(&h);
            for (int i = 0; i < 64; i++) b[i] = (unsigned char)g[(size_t)r * 64 + i];
            void *ptr = nullptr;
            if (hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); mapped = false; break; }
            d->pw_peer_rx[(size_t)p] = (unsigned char *)ptr;
            // the peer's slot size follows from ITS halo length, which this rank does not know: the peer's flag block
            // therefore sits at a place both sides can compute — see flag_base below
        }
    }
    // the flag block's offset inside a peer's buffer depends on the peer's halo length: exchange it too
    std::vector<double> my_flag((size_t)np, (double)flag_off), their_flag((size_t)np, 0.), my_slot((size_t)np, (double)slot_bytes), their_slot((size_t)np, 0.);
    {
        std::vector<const double *> sp((size_t)np);
        std::vector<double *> rp((size_t)np);
        std::vector<int64_t> one((size_t)np, 1);
        for (int p = 0; p < np; p++) { sp[(size_t)p] = &my_flag[(size_t)p]; rp[(size_t)p] = &their_flag[(size_t)p]; }
        MGCR_TRY(comm_exchange_host(c, np, P->peers.data(), sp.data(), one.data(), rp.data(), one.data()));
        for (int p = 0; p < np; p++) { sp[(size_t)p] = &my_slot[(size_t)p]; rp[(size_t)p] = &their_slot[(size_t)p]; }
        MGCR_TRY(comm_exchange_host(c, np, P->peers.data(), sp.data(), one.data(), rp.data(), one.data()));
    }
    if (mapped) {
        for (int p = 0; p < np; p++) {
            const int r = P->peers[(size_t)p];
            unsigned char *rb = d->pw_peer_rx[(size_t)p];
            HaloPwPeer &e = tab[(size_t)p];
            for (int sl = 0; sl < 2; sl++) {
                e.dst[sl] = reinterpret_cast<cplx *>(rb + (size_t)sl * (size_t)their_slot[(size_t)p]) + (int64_t)their_off[(size_t)p];
                e.flag_remote[sl] = reinterpret_cast<uint64_t *>(rb + (size_t)their_flag[(size_t)p]) + (size_t)sl * PW_MAX_RANKS + (size_t)c->rank;
                e.flag_local[sl] = reinterpret_cast<const uint64_t *>(d->pw_rx + flag_off) + (size_t)sl * PW_MAX_RANKS + (size_t)r;
            }
            e.send_off = d->send_off[(size_t)p];
            e.send_cnt = d->send_cnt[(size_t)p];
            for (int sl = 0; sl < 2; sl++) e.rx_local[sl] = reinterpret_cast<cplx *>(d->pw_rx + (size_t)sl * slot_bytes) + P->recv_off[(size_t)p];
            e.recv_cnt = P->recv_count[(size_t)p];
            max_cnt = std::max(max_cnt, e.send_cnt);
        }
        d->pw_grid_x = (unsigned)std::min<int64_t>(std::max<int64_t>((max_cnt + 255) / 256, 1), 1024);
        mapped = np == 0 || (hipMalloc((void **)&d->pw_tab, sizeof(HaloPwPeer) * (size_t)np) == hipSuccess &&
                             hipMemcpy(d->pw_tab, tab.data(), sizeof(HaloPwPeer) * (size_t)np, hipMemcpyHostToDevice) == hipSuccess);
        if (!mapped) (void)hipGetLastError();
    }
    double flag = mapped ? 0. : 1.;
    MGCR_TRY(comm_allreduce_host(c, &flag, 1));   // nobody stores into a slot before everybody has mapped its neighbours
    good = good && flag == 0.;
    if (good && np > 0) {
        // self-test through both slots: x holds the global row numbers, the halo must then hold halo_gid
        cplx *xt = nullptr;
        good = hipMalloc((void **)&xt, sizeof(cplx) * (size_t)std::max<int64_t>(P->nloc, 1)) == hipSuccess;
        std::vector<cplx> got(nh);
        for (int t = 0; t < 2 && good; t++) {
            const double im = 0.5 + t;
            if (P->nloc) hipLaunchKernelGGL(halo_test_fill_kernel, dim3((unsigned)((P->nloc + 255) / 256)), dim3(256), 0, ctx().stream, xt, P->nloc, P->row0, im);
            good = halo_pw_launch(d, xt) == MGCR_OK && hipStreamSynchronize(ctx().stream) == hipSuccess;
            if (good && nh) good = hipMemcpy(got.data(), halo_pw_slot(d, d->pw_seq), sizeof(cplx) * nh, hipMemcpyDeviceToHost) == hipSuccess;
            for (size_t j = 0; j < nh && good; j++) good = got[j].x == (double)P->halo_gid[j] && got[j].y == im;
            if (good) good = *(volatile int *)c->pw_err == 0;
        }
        if (!good) (void)hipGetLastError();
        if (xt) hipFree(xt);
    }
    flag = good ? 0. : 1.;
    MGCR_TRY(comm_allreduce_host(c, &flag, 1));
    if (flag == 0.) {
        d->pw_on = np > 0;
        if (!d->pw_on) halo_pw_release(d);
    } else {
        hipDeviceSynchronize();
        *(volatile int *)c->pw_err = 0;
        halo_pw_release(d);
        (void)hipGetLastError();
    }
    return MGCR_OK;
}

// the halo segment the exchange begun last delivers into (call after dist_halo_begin)
const cplx *dist_halo_ptr(DistCsr *d) { return d->pw_on ? halo_pw_slot(d, d->pw_seq) : d->xh; }

int dist_halo_begin(DistCsr *d, const cplx *x, bool overlap_interior) {
    Comm *c = d->comm;
    Plan *P = d->plan;
    const int np = (int)P->peers.size();
    if (np == 0) return MGCR_OK;
    if (d->pw_on) return halo_pw_launch(d, x, overlap_interior);
    hipStream_t main = ctx().stream;
    int64_t tot_send = d->send_off.empty() ? 0 : d->send_off.back() + d->send_cnt.back();
    // pack the non-contiguous send lists
    for (int p = 0; p < np; p++)
        if (d->send_contig[(size_t)p] < 0 && d->send_cnt[(size_t)p]) {
            hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((d->send_cnt[(size_t)p] + 255) / 256)), dim3(256), 0, main,
                               d->send_cnt[(size_t)p], d->send_idx + d->send_off[(size_t)p], x, d->sendbuf + d->send_off[(size_t)p],
                               get_apply_skip().p, get_apply_skip().it);
            MGCR_HIP(hipGetLastError());
        }
    if (c->is_rccl) {
        // Default: the exchange is ordered on the compute stream, like the all-reduces — every RCCL
        // call of this communicator then sits on one stream, in the same order on every rank.
        // MGCR_HALO_OVERLAP=1 moves it to the communication stream so that it overlaps the interior
        // rows (to be switched on once it has been exercised on a multi-GPU node).
        hipStream_t cs = halo_overlap() ? c->comm_stream : main;
        if (cs != main) {
            MGCR_HIP(hipEventRecord(c->ev_ready, main));
            MGCR_HIP(hipStreamWaitEvent(cs, c->ev_ready, 0));
        }
        MGCR_NCCL(rccl().GroupStart());
        for (int p = 0; p < np; p++) {
            const cplx *src = d->send_contig[(size_t)p] >= 0 ? x + d->send_contig[(size_t)p] : d->sendbuf + d->send_off[(size_t)p];
            if (d->send_cnt[(size_t)p])
                MGCR_NCCL(rccl().Send(src, (size_t)d->send_cnt[(size_t)p] * 2, ncclDouble, P->peers[(size_t)p], c->nccl, cs));
            if (P->recv_count[(size_t)p])
                MGCR_NCCL(rccl().Recv(d->xh + P->recv_off[(size_t)p], (size_t)P->recv_count[(size_t)p] * 2, ncclDouble, P->peers[(size_t)p], c->nccl, cs));
        }
        MGCR_NCCL(rccl().GroupEnd());
        if (cs != main) MGCR_HIP(hipEventRecord(c->ev_done, cs));
        return MGCR_OK;
    }
    // host-staged transport: device -> host, callback, host -> device (synchronous)
    d->h_send.resize((size_t)tot_send * 2);
    d->h_recv.resize(P->halo_gid.size() * 2);
    for (int p = 0; p < np; p++) {
        const cplx *src = d->send_contig[(size_t)p] >= 0 ? x + d->send_contig[(size_t)p] : d->sendbuf + d->send_off[(size_t)p];
        if (d->send_cnt[(size_t)p])
            MGCR_HIP(hipMemcpyAsync(d->h_send.data() + 2 * d->send_off[(size_t)p], src, sizeof(cplx) * (size_t)d->send_cnt[(size_t)p], hipMemcpyDeviceToHost, main));
    }
    MGCR_HIP(hipStreamSynchronize(main));
    std::vector<const double *> sp((size_t)np);
    std::vector<double *> rp((size_t)np);
    std::vector<int64_t> sc((size_t)np), rcv((size_t)np);
    for (int p = 0; p < np; p++) {
        sp[(size_t)p] = d->h_send.data() + 2 * d->send_off[(size_t)p];
        sc[(size_t)p] = 2 * d->send_cnt[(size_t)p];
        rp[(size_t)p] = d->h_recv.data() + 2 * P->recv_off[(size_t)p];
        rcv[(size_t)p] = 2 * P->recv_count[(size_t)p];
    }
    MGCR_TRY(comm_exchange_host(c, np, P->peers.data(), sp.data(), sc.data(), rp.data(), rcv.data()));
    if (!P->halo_gid.empty())
        MGCR_HIP(hipMemcpyAsync(d->xh, d->h_recv.data(), sizeof(cplx) * P->halo_gid.size(), hipMemcpyHostToDevice, main));
    return MGCR_OK;
}

int dist_halo_end(DistCsr *d) {
    Comm *c = d->comm;
    if (d->pw_wait_pending) {   // second half of a split peer-write exchange (halo_pw_kernel<false>)
        d->pw_wait_pending = false;
        const int np = (int)d->plan->peers.size();
        hipLaunchKernelGGL(halo_pw_wait_kernel, dim3(1), dim3(64), 0, ctx().stream, (const HaloPwPeer *)d->pw_tab, np, d->pw_seq, c->pw_err, pw_timeout_run());
        MGCR_HIP(hipGetLastError());
        g_halo_split_count++;
        return MGCR_OK;
    }
    if (c->is_rccl && halo_overlap() && !d->plan->peers.empty()) MGCR_HIP(hipStreamWaitEvent(ctx().stream, c->ev_done, 0));
    return MGCR_OK;
}

void dist_info(DistCsr *d, const cplx **xh, int64_t *interior_begin, int64_t *interior_end) {
    *xh = dist_halo_ptr(d);
    *interior_begin = d->plan->interior_begin;
    *interior_end = d->plan->interior_end;
}

Comm *dist_comm(DistCsr *d) { return d->comm; }

void dist_sizes(DistCsr *d, int64_t *nloc, int64_t *nh, int64_t *row0, int64_t *n_global, int *rank, int *nranks) {
    if (nloc) *nloc = d->plan->nloc;
    if (nh) *nh = (int64_t)d->plan->halo_gid.size();
    if (row0) *row0 = d->plan->row0;
    if (n_global) *n_global = d->plan->n_global;
    if (rank) *rank = d->comm->rank;
    if (nranks) *nranks = d->comm->nranks;
}

int comm_allreduce_host_pub(Comm *c, double *buf, int64_t count) { return comm_allreduce_host(c, buf, count); }

// host-level halo exchange of w doubles per row (set-up data: aggregate ids, prolongator rows):
// own[nloc*w] -> halo[nh*w], same lists as the SpMV halo
int dist_exchange_rows_host(DistCsr *d, const double *own, int w, double *halo) {
    Plan *P = d->plan;
    const int np = (int)P->peers.size();
    std::vector<std::vector<double>> sb((size_t)np);
    std::vector<const double *> sp((size_t)np);
    std::vector<double *> rp((size_t)np);
    std::vector<int64_t> sc((size_t)np), rc((size_t)np);
    for (int p = 0; p < np; p++) {
        const std::vector<int64_t> &rows = P->send_rows[(size_t)p];
        sb[(size_t)p].resize(rows.size() * (size_t)w);
        for (size_t i = 0; i < rows.size(); i++)
            memcpy(sb[(size_t)p].data() + i * (size_t)w, own + (size_t)rows[i] * (size_t)w, sizeof(double) * (size_t)w);
        sp[(size_t)p] = sb[(size_t)p].data();
        sc[(size_t)p] = (int64_t)rows.size() * w;
        rp[(size_t)p] = halo + (size_t)P->recv_off[(size_t)p] * (size_t)w;
        rc[(size_t)p] = P->recv_count[(size_t)p] * w;
    }
    return comm_exchange_host(d->comm, np, P->peers.data(), sp.data(), sc.data(), rp.data(), rc.data());
}

void dist_free(DistCsr *d) {
    if (!d) return;
    halo_pw_release(d);
    hipFree(d->xh); hipFree(d->sendbuf); hipFree(d->send_idx);
    delete d->plan;
    delete d;
}

int comm_nranks(Comm *c) { return c ? c->nranks : 1; }

// does a solve on this communicator go through the fold + all-reduce path?  (MGCR_TEST_FORCE_COLLECTIVES=1
// turns it on for a 1-rank communicator too, so that the RCCL calls can be exercised on a single GPU)
bool comm_collectives(Comm *c) {
    if (!c) return false;
    if (c->nranks > 1) return true;
    static const bool force = getenv("MGCR_TEST_FORCE_COLLECTIVES") && atoi(getenv("MGCR_TEST_FORCE_COLLECTIVES")) != 0;
    return force;
}

// in-place sum over ranks of `count` doubles in device memory, ordered on the compute stream
int comm_allreduce_dev(Comm *c, double *dbuf, int count) {
    if (!c || !comm_collectives(c)) return MGCR_OK;
    hipStream_t st = ctx().stream;
    if (c->pw_on && count <= PW_MAX_SCALARS) return pw_launch(c, nullptr, count, nullptr, 0, dbuf, 0);
    if (c->is_rccl) {
        MGCR_NCCL(rccl().AllReduce(dbuf, dbuf, (size_t)count, ncclDouble, ncclSum, c->nccl, st));
        return MGCR_OK;
    }
    MGCR_TRY(comm_device_ready(c));
    MGCR_CHECK(count <= 1024, MGCR_ERR_UNSUPPORTED, "all-reduce of %d scalars exceeds the staging buffer", count);
    MGCR_HIP(hipMemcpyAsync(c->h_pin, dbuf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, st));
    MGCR_HIP(hipStreamSynchronize(st));
    MGCR_TRY(comm_allreduce_host(c, c->h_pin, count));
    MGCR_HIP(hipMemcpyAsync(dbuf, c->h_pin, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, st));
    MGCR_HIP(hipStreamSynchronize(st));  // h_pin is reused by the next call
    return MGCR_OK;
}

// out[0..na) = sum over ranks of the folded partials pa, out[na..na+nb) likewise of pb (slabs of RED_MAX_BLOCKS per
// scalar, nblk workgroup partials each; blas1.hip k_fold2): one kernel when the peer-write all-reduce is up
int comm_fold_allreduce(Comm *c, const double *pa, int na, const double *pb, int nb, double *out, int nblk) {
    if (c && c->pw_on && comm_collectives(c) && na + nb <= PW_MAX_SCALARS) return pw_launch(c, pa, na, pb, nb, out, nblk);
    MGCR_TRY(k_fold2(pa, na, out, pb, nb, out + na, nblk));
    return comm_allreduce_dev(c, out, na + nb);
}

// did a peer-write wait time out since the last check?  (called where a distributed solve hands back to the host)
int comm_check(Comm *c) {
    if (c && c->pw_err && *(volatile int *)c->pw_err != 0) {
        *(volatile int *)c->pw_err = 0;
        set_error("peer-write exchange (all-reduce or halo): a rank did not arrive within the time limit; the communicator is no longer usable");
        return MGCR_ERR_COMM;
    }
    return MGCR_OK;
}

int comm_live_count() {
    std::lock_guard<std::mutex> lk(live_comms_mtx());
    return (int)live_comms().size();
}

int comm_check_all() {
    std::lock_guard<std::mutex> lk(live_comms_mtx());
    int rc = MGCR_OK;
    for (Comm *c : live_comms()) {
        int r = comm_check(c);
        if (r != MGCR_OK) rc = r;
    }
    return rc;
}

int dist_halo_kind(DistCsr *d) { return d->pw_on ? 2 : d->comm->is_rccl ? 1 : 0; }
int comm_allreduce_kind(Comm *c) { return c->pw_on ? 2 : c->is_rccl ? 1 : 0; }

// the device side of a partition plan: halo segment, send lists, peer-write receive slots (collective: the peer-write
// self-tests run here).  Takes ownership of P.
static int dist_attach(Comm *c, Plan *P, DistCsr **out) {
    DistCsr *d = new DistCsr();
    d->comm = c;
    d->plan = P;
    d->pw_seq = pw_seq0();
    const int64_t nh = (int64_t)P->halo_gid.size();
    const int np = (int)P->peers.size();
    int64_t off = 0;
    std::vector<int32_t> idx;
    for (int p = 0; p < np; p++) {
        const std::vector<int64_t> &rows = P->send_rows[(size_t)p];
        d->send_off.push_back(off);
        d->send_cnt.push_back((int64_t)rows.size());
        bool contig = !rows.empty();
        for (size_t i = 1; i < rows.size() && contig; i++) contig = rows[i] == rows[i - 1] + 1;
        d->send_contig.push_back(contig ? rows[0] : -1);
        for (int64_t r : rows) idx.push_back((int32_t)r);
        off += (int64_t)rows.size();
    }
    hipError_t e = hipSuccess;
    if (nh) e = hipMalloc((void **)&d->xh, sizeof(cplx) * (size_t)nh);
    if (e == hipSuccess && off) e = hipMalloc((void **)&d->sendbuf, sizeof(cplx) * (size_t)off);
    if (e == hipSuccess && off) e = hipMalloc((void **)&d->send_idx, sizeof(int32_t) * (size_t)off);
    if (e == hipSuccess && off) e = hipMemcpy(d->send_idx, idx.data(), sizeof(int32_t) * (size_t)off, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dist_free(d);
        set_error("distributed operator: device allocation failed: %s", hipGetErrorString(e));
        return MGCR_ERR_ALLOC;
    }
    int rc = comm_device_ready(c);
    if (rc == MGCR_OK) rc = comm_pw_setup(c);
    if (rc == MGCR_OK) rc = halo_pw_setup(d);
    if (rc != MGCR_OK) { dist_free(d); return rc; }
    *out = d;
    return MGCR_OK;
}

int dist_csr_create(Comm *c, int64_t n_global, int64_t row0, int64_t nloc, const int64_t *rowptr, const int64_t *col,
                    const double *val_ri, Op *op) {
    Plan *P = nullptr;
    MGCR_TRY(plan_build(c, n_global, row0, nloc, rowptr, col, &P));
    const int64_t nh = (int64_t)P->halo_gid.size();
    int rc = csr_build_device(nloc, nloc + nh, rowptr, P->col_local.data(), val_ri, &op->csr);
    if (rc != MGCR_OK) { delete P; return rc; }
    std::vector<int64_t>().swap(P->col_local);
    DistCsr *d = nullptr;
    rc = dist_attach(c, P, &d);
    if (rc != MGCR_OK) { csr_free(&op->csr); return rc; }
    op->dist = d;
    op->comm = c;
    return MGCR_OK;
}

// Row block of a distributed HierarchicalSparse (src/HierarchicalSparse.h:101-161): block rows [brow0, brow0 + nbloc) of
// nb_global, block columns GLOBAL.  The partition plan is made at BLOCK granularity (a halo entry = one block row of x,
// bs values) and expanded to the element lists the halo machinery above works on; the apply reads a block column's bs
// values from x (owned) or in place from the halo segment.
int dist_bcsr_create(Comm *c, int64_t nb_global, int64_t brow0, int32_t nbloc, int32_t bs, const int32_t *browptr,
                     const int64_t *bcol_global, const double *blocks_ri, Op *op) {
    MGCR_CHECK(bs >= 1 && nbloc >= 0 && browptr && browptr[0] == 0, MGCR_ERR_INVALID, "dist_bcsr_create: bad argument");
    std::vector<int64_t> rp((size_t)nbloc + 1);
    for (int32_t r = 0; r <= nbloc; r++) rp[(size_t)r] = browptr[r];
    Plan *B = nullptr;
    MGCR_TRY(plan_build(c, nb_global, brow0, nbloc, rp.data(), bcol_global, &B));
    const int64_t nhb = (int64_t)B->halo_gid.size();
    MGCR_CHECK(((int64_t)nbloc + nhb) * bs < ((int64_t)1 << 31), MGCR_ERR_UNSUPPORTED, "row block too large");
    std::vector<int32_t> bcol_local((size_t)B->nnz);
    for (int64_t l = 0; l < B->nnz; l++) bcol_local[(size_t)l] = (int32_t)B->col_local[(size_t)l];
    int rc = bcsr_build_device(nbloc, (int32_t)(nbloc + nhb), bs, browptr, bcol_local.data(), blocks_ri, &op->bcsr);
    if (rc != MGCR_OK) { delete B; return rc; }
    // element-level plan: block b -> elements b*bs .. b*bs + bs - 1, same order
    Plan *P = new Plan();
    P->comm = c; P->n_global = nb_global * bs; P->row0 = brow0 * bs; P->nloc = (int64_t)nbloc * bs; P->nnz = 0;
    for (int64_t o : B->offsets) P->offsets.push_back(o * bs);
    for (int64_t g : B->halo_gid) for (int32_t k = 0; k < bs; k++) P->halo_gid.push_back(g * bs + k);
    P->peers = B->peers;
    for (size_t p = 0; p < B->peers.size(); p++) {
        P->recv_count.push_back(B->recv_count[p] * bs);
        P->recv_off.push_back(B->recv_off[p] * bs);
        std::vector<int64_t> rows;
        for (int64_t r : B->send_rows[p]) for (int32_t k = 0; k < bs; k++) rows.push_back(r * bs + k);
        P->send_rows.push_back(rows);
    }
    P->interior_begin = B->interior_begin * bs;
    P->interior_end = B->interior_end * bs;
    delete B;
    DistCsr *d = nullptr;
    rc = dist_attach(c, P, &d);
    if (rc != MGCR_OK) { bcsr_free(&op->bcsr); return rc; }
    op->dist = d;
    op->comm = c;
    return MGCR_OK;
}

}  // namespace mgcr

using namespace mgcr;
struct mgcr_comm_s : mgcr::Comm {};
struct mgcr_plan_s : mgcr::Plan {};

#define LOCK() std::lock_guard<std::recursive_mutex> lk__(ctx().mtx)

extern "C" {

int mgcr_rccl_unique_id(void *id128) {
    MGCR_CHECK(id128, MGCR_ERR_INVALID, "null id buffer");
    MGCR_TRY(rccl_load());
    ncclUniqueId id;
    MGCR_NCCL(rccl().GetUniqueId(&id));
    static_assert(sizeof(id) == MGCR_RCCL_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return MGCR_OK;
}

int mgcr_comm_create_rccl(int rank, int nranks, const void *id128, mgcr_comm_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(out && id128 && nranks >= 1 && rank >= 0 && rank < nranks, MGCR_ERR_INVALID, "mgcr_comm_create_rccl: bad argument");
    MGCR_TRY(rccl_load());
    LOCK();
    mgcr_comm_s *c = new mgcr_comm_s();
    c->rank = rank; c->nranks = nranks; c->is_rccl = true;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = rccl().CommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank failed: %s", rccl().GetErrorString(r));
        delete c;
        return MGCR_ERR_COMM;
    }
    int rc = comm_device_ready(c);
    if (rc != MGCR_OK) { delete c; return rc; }
    c->pw_seq = pw_seq0();
    comm_register(c);
    *out = c;
    return MGCR_OK;
}

int mgcr_comm_create_host(int rank, int nranks, mgcr_allreduce_cb allreduce, mgcr_exchange_cb exchange, void *user, mgcr_comm_t *out) {
    MGCR_CHECK(out && nranks >= 1 && rank >= 0 && rank < nranks && (nranks == 1 || (allreduce && exchange)), MGCR_ERR_INVALID,
               "mgcr_comm_create_host: bad argument");
    mgcr_comm_s *c = new mgcr_comm_s();
    c->rank = rank; c->nranks = nranks; c->is_rccl = false;
    c->allreduce = allreduce; c->exchange = exchange; c->user = user;
    c->pw_seq = pw_seq0();
    comm_register(c);
    *out = c;
    return MGCR_OK;
}

int mgcr_comm_allreduce_sum(mgcr_comm_t c, double *buf, int32_t count) {
    MGCR_CHECK(c && buf && count > 0, MGCR_ERR_INVALID, "mgcr_comm_allreduce_sum: bad argument");
    return comm_allreduce_host(c, buf, count);
}

int mgcr_comm_allreduce_kind(mgcr_comm_t c, int32_t *kind) {
    MGCR_CHECK(c && kind, MGCR_ERR_INVALID, "mgcr_comm_allreduce_kind: null argument");
    *kind = comm_allreduce_kind(c);
    return MGCR_OK;
}

int mgcr_comm_bench_allreduce(mgcr_comm_t c, int32_t count, int32_t reps, double *us_avg) {
    MGCR_CHECK(c && count > 0 && count <= 64 && reps > 0 && us_avg, MGCR_ERR_INVALID, "mgcr_comm_bench_allreduce: bad argument");
    MGCR_TRY(require_ctx());
    MGCR_TRY(comm_pw_setup(c));
    Context &cx = ctx();
    double *d = nullptr;
    MGCR_HIP(hipMalloc((void **)&d, sizeof(double) * 64));
    MGCR_HIP(hipMemsetAsync(d, 0, sizeof(double) * 64, cx.stream));
    int rc = comm_allreduce_dev(c, d, count);  // warm-up
    MGCR_HIP(hipEventRecord(cx.ev0, cx.stream));
    for (int i = 0; i < reps && rc == MGCR_OK; i++) rc = comm_allreduce_dev(c, d, count);
    MGCR_HIP(hipEventRecord(cx.ev1, cx.stream));
    MGCR_HIP(hipEventSynchronize(cx.ev1));
    float ms = 0.f;
    hipEventElapsedTime(&ms, cx.ev0, cx.ev1);
    hipFree(d);
    *us_avg = 1e3 * (double)ms / reps;
    if (rc == MGCR_OK) rc = comm_check(c);
    return rc;
}

int mgcr_comm_destroy(mgcr_comm_t c) {
    if (!c) return MGCR_OK;
    comm_unregister(c);
    if (ctx().ready) {
        hipStreamSynchronize(ctx().stream);
        if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    }
    pw_release(c);
    if (c->is_rccl && c->nccl) rccl().CommDestroy(c->nccl);
    if (c->comm_stream) hipStreamDestroy(c->comm_stream);
    if (c->ev_ready) hipEventDestroy(c->ev_ready);
    if (c->ev_done) hipEventDestroy(c->ev_done);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->h_pin) hipHostFree(c->h_pin);
    delete c;
    return MGCR_OK;
}

int mgcr_plan_create(mgcr_comm_t comm, int64_t n_global, int64_t row0, int64_t nrow_local, const int64_t *rowptr,
                     const int64_t *col_global, mgcr_plan_t *out) {
    MGCR_CHECK(comm && out && rowptr, MGCR_ERR_INVALID, "mgcr_plan_create: null argument");
    Plan *P = nullptr;
    MGCR_TRY(plan_build(comm, n_global, row0, nrow_local, rowptr, col_global, &P));
    *out = static_cast<mgcr_plan_s *>(P);
    return MGCR_OK;
}

int mgcr_plan_info(mgcr_plan_t plan, int64_t *n_halo, int32_t *npeers, int64_t *interior_begin, int64_t *interior_end) {
    MGCR_CHECK(plan, MGCR_ERR_INVALID, "null plan");
    if (n_halo) *n_halo = (int64_t)plan->halo_gid.size();
    if (npeers) *npeers = (int32_t)plan->peers.size();
    if (interior_begin) *interior_begin = plan->interior_begin;
    if (interior_end) *interior_end = plan->interior_end;
    return MGCR_OK;
}

int mgcr_plan_peers(mgcr_plan_t plan, int32_t *peers, int64_t *send_counts, int64_t *recv_counts) {
    MGCR_CHECK(plan, MGCR_ERR_INVALID, "null plan");
    for (size_t p = 0; p < plan->peers.size(); p++) {
        if (peers) peers[p] = plan->peers[p];
        if (send_counts) send_counts[p] = (int64_t)plan->send_rows[p].size();
        if (recv_counts) recv_counts[p] = plan->recv_count[p];
    }
    return MGCR_OK;
}

int mgcr_plan_local_columns(mgcr_plan_t plan, int64_t *col_local) {
    MGCR_CHECK(plan && col_local, MGCR_ERR_INVALID, "null argument");
    MGCR_CHECK((int64_t)plan->col_local.size() == plan->nnz, MGCR_ERR_INVALID, "plan no longer holds its column map");
    memcpy(col_local, plan->col_local.data(), sizeof(int64_t) * (size_t)plan->nnz);
    return MGCR_OK;
}

int mgcr_plan_send_indices(mgcr_plan_t plan, int32_t peer_slot, int64_t *local_rows) {
    MGCR_CHECK(plan && local_rows && peer_slot >= 0 && peer_slot < (int32_t)plan->peers.size(), MGCR_ERR_INVALID, "bad argument");
    const std::vector<int64_t> &r = plan->send_rows[(size_t)peer_slot];
    memcpy(local_rows, r.data(), sizeof(int64_t) * r.size());
    return MGCR_OK;
}

int mgcr_plan_halo_globals(mgcr_plan_t plan, int64_t *global_cols) {
    MGCR_CHECK(plan && global_cols, MGCR_ERR_INVALID, "null argument");
    memcpy(global_cols, plan->halo_gid.data(), sizeof(int64_t) * plan->halo_gid.size());
    return MGCR_OK;
}

int mgcr_plan_destroy(mgcr_plan_t plan) {
    delete static_cast<mgcr::Plan *>(plan);
    return MGCR_OK;
}

int mgcr_dbcsr_create(mgcr_comm_t comm, int64_t nb_global, int64_t brow0, int32_t nbrow_local, int32_t bs, const int32_t *browptr,
                      const int64_t *bcol_global, const double *blocks_ri, mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(comm && out && browptr && (bcol_global || browptr[nbrow_local] == 0) && (blocks_ri || browptr[nbrow_local] == 0),
               MGCR_ERR_INVALID, "mgcr_dbcsr_create: null argument");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_BCSR;
    op->dim = (int64_t)nbrow_local * bs;      // Fields of a distributed operator hold this rank's rows
    op->nrow = (int64_t)nbrow_local * bs;
    int rc = dist_bcsr_create(comm, nb_global, brow0, nbrow_local, bs, browptr, bcol_global, blocks_ri, op);
    if (rc != MGCR_OK) { delete op; return rc; }
    *out = op;
    return MGCR_OK;
}

int mgcr_dcsr_create(mgcr_comm_t comm, int64_t n_global, int64_t row0, int64_t nrow_local, const int64_t *rowptr,
                     const int64_t *col_global, const double *val_ri, mgcr_op_t *out) {
    MGCR_TRY(require_ctx());
    MGCR_CHECK(comm && out && rowptr, MGCR_ERR_INVALID, "mgcr_dcsr_create: null argument");
    LOCK();
    mgcr_op_s *op = new mgcr_op_s();
    op->kind = OP_CSR;
    op->dim = nrow_local;
    op->nrow = nrow_local;
    int rc = dist_csr_create(comm, n_global, row0, nrow_local, rowptr, col_global, val_ri, op);
    if (rc != MGCR_OK) { delete op; return rc; }
    *out = op;
    return MGCR_OK;
}

}  // extern "C"
