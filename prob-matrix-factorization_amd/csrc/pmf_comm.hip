// Multi-GPU side of libpmf_hip.so (SURVEY.md section 8(b) `pmf_comm_init`, section 8(e)).
//
// Ratings are sharded by user range, one process per GPU; the item block is replicated.  An item
// half-sweep is  accumulate (raw per-item sums over this rank's ratings) -> all-reduce -> finalize.
// This file owns everything around the collective: the RCCL communicator, a second (high-priority)
// HIP stream the collectives run on, the event ordering between that stream and the context's
// compute stream, the library-owned statistics buffers, and the row-chunk pipeline that lets the
// all-reduce of chunk c travel over xGMI while chunk c+1 is being accumulated.
//
// Two exchanges of a chunk's statistics (pmf_comm_set_exchange):
//   ALLREDUCE        ncclAllReduce, then EVERY rank finalises every row of the chunk;
//   SCATTER_GATHER   ncclReduceScatter -> each rank finalises only ITS 1/N of the chunk's rows -> ncclAllGather of
//                    the finalised state.  Same wire bytes for the Gaussian model (statistics and state are both
//                    [Kp + Kpad] per row), 1/N of the K x K row solves per rank; every rank receives the same
//                    finalised bytes, so the replicas stay bit-identical.
//
// The product library has ONE transport, RCCL (ncclAllReduce / ncclReduceScatter / ncclAllGather / ncclBroadcast on
// the collective stream, one rank per GPU).  Built with -DPMF_TEST_TRANSPORT (libpmf_hip_test.so, tests only) it also
// carries HOSTSHM: ranks of ONE GPU exchange through POSIX shared memory in stream order (hipLaunchHostFunc).  RCCL
// refuses two ranks on one device, so this is how the multi-rank code path -- chunk offsets, event ordering, the
// sharded fit, both exchanges -- is rehearsed on a one-GPU box.  Slow, node-local, never in the product build.
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#ifdef PMF_TEST_TRANSPORT
#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#endif

#include <algorithm>
#include <atomic>
#include <new>

#include "pmf_internal.h"

#define PMF_NCCL_CHECK(expr)                                                                  \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            pmf_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
            return PMF_ECOMM;                                                                 \
        }                                                                                     \
    } while (0)

namespace {

constexpr size_t kSmallBytes = 64u << 10;  // device / pinned buffer of the host-value collectives
#ifdef PMF_TEST_TRANSPORT
constexpr size_t kShmHeader = 4096;
constexpr size_t kShmStage = 32u << 20;    // staging bytes per rank (messages travel in pieces of this size)
constexpr double kShmTimeoutS = 120.0;     // default deadline of a hostshm rendezvous / barrier (PMF_COMM_TIMEOUT_S overrides)

struct ShmHeader {
    std::atomic<int> count;
    std::atomic<int> generation;
    std::atomic<int> attached;
    std::atomic<int> failed;
};
#endif

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

}  // namespace

struct PmfComm {
    int refs = 0;
    int nranks = 1, rank = 0, device = 0;
    int transport = PMF_TRANSPORT_RCCL;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr;             // the collectives' stream
    hipStream_t fin_stream = nullptr;         // finalize(c) runs here, beside the accumulation of later chunks
    std::vector<hipEvent_t> ev_ready, ev_done, ev_fin, ev_gath;  // per row chunk: statistics ready / reduced / finalized / state gathered
    void *d_small = nullptr;                  // kSmallBytes, host-value collectives
    void *h_small = nullptr;                  // pinned twin
    bool failed = false;                      // a collective failed or timed out: every later call returns PMF_ECOMM
    double timeout_s = 1800.0;                // PMF_COMM_TIMEOUT_S
    bool timeout_from_env = false;
#ifdef PMF_TEST_TRANSPORT
    // HOSTSHM
    char shm_name[64] = "";
    void *shm = nullptr;
    size_t shm_bytes = 0;
    bool shm_registered = false;
    void *h_result = nullptr;                 // pinned, kShmStage: this rank's reduced piece
    ShmHeader *hdr() const { return (ShmHeader *)shm; }
    char *stage(int r) const { return (char *)shm + kShmHeader + (size_t)r * kShmStage; }
    // deadline of a rendezvous / barrier: PMF_COMM_TIMEOUT_S when the environment sets it (a rank whose accumulate
    // lags on a shared GPU must not poison the communicator at a fixed 120 s), else 120 s
    double shm_deadline() const { return timeout_from_env && timeout_s > 0 ? timeout_s : kShmTimeoutS; }
    bool peer_failed() const { return transport == PMF_TRANSPORT_HOSTSHM && shm && hdr()->failed.load(); }
#else
    bool peer_failed() const { return false; }
#endif
};

#ifdef PMF_TEST_TRANSPORT
// ---------------------------------------------------------------------------
// HOSTSHM transport (test builds only)
// ---------------------------------------------------------------------------
namespace {

// Sense-reversing barrier over the shared header.  Returns false on timeout / a failed peer.
bool shm_barrier(PmfComm *cm) {
    ShmHeader *h = cm->hdr();
    if (h->failed.load()) return false;
    const int gen = h->generation.load();
    if (h->count.fetch_add(1) + 1 == cm->nranks) {
        h->count.store(0);
        h->generation.fetch_add(1);
        return true;
    }
    const double t0 = now_s();
    int spins = 0;
    while (h->generation.load() == gen) {
        if (h->failed.load()) return false;
        if ((++spins & 1023) == 0) {
            if (now_s() - t0 > cm->shm_deadline()) {
                h->failed.store(1);
                return false;
            }
            sched_yield();
        }
    }
    return true;
}

struct ShmOp {
    PmfComm *cm;
    size_t bytes;    // bytes of the staged piece
    size_t lo, hi;   // the part of the piece THIS rank reduces and takes back (all of it in an all-reduce; its
                     // intersection with the rank's own slice in a reduce-scatter)
    int dtype;   // PMF_F32 / PMF_F64; -1 = broadcast
    int op;      // PMF_OP_SUM / PMF_OP_MAX
    int root;
};

template <typename T>
void shm_reduce(PmfComm *cm, size_t lo, size_t hi, int op) {   // element range [lo, hi) of the staged piece
    T *out = (T *)cm->h_result;
    const T *first = (const T *)cm->stage(0);
    for (size_t k = lo; k < hi; ++k) out[k] = first[k];
    for (int r = 1; r < cm->nranks; ++r) {   // rank order: every rank gets bit-identical sums
        const T *src = (const T *)cm->stage(r);
        if (op == PMF_OP_MAX)
            for (size_t k = lo; k < hi; ++k) out[k] = src[k] > out[k] ? src[k] : out[k];
        else
            for (size_t k = lo; k < hi; ++k) out[k] += src[k];
    }
}

// runs on the collective stream's callback thread, after this rank's D2H copy of the piece
void shm_host_step(void *arg) {
    ShmOp *o = (ShmOp *)arg;
    PmfComm *cm = o->cm;
    if (shm_barrier(cm)) {                     // every rank's piece is staged
        if (o->dtype < 0) memcpy(cm->h_result, cm->stage(o->root), o->bytes);
        else if (o->dtype == PMF_F64) shm_reduce<double>(cm, o->lo / 8, o->hi / 8, o->op);
        else shm_reduce<float>(cm, o->lo / 4, o->hi / 4, o->op);
        (void)shm_barrier(cm);                 // nobody overwrites a staging area that is still being read
    }
    delete o;
}

// `own_lo`, `own_hi`: the byte range of the message this rank keeps (reduce-scatter); the whole message otherwise
int shm_collective(PmfComm *cm, const void *send, void *recv, size_t bytes, int dtype, int op, int root,
                   size_t own_lo = 0, size_t own_hi = (size_t)-1) {
    PMF_REQUIRE(!cm->hdr()->failed.load(), PMF_ECOMM, "hostshm transport: a peer failed or timed out");
    const bool bcast = dtype < 0;
    own_hi = std::min(own_hi, bytes);
    for (size_t off = 0; off < bytes; off += kShmStage) {
        const size_t n = std::min(kShmStage, bytes - off);
        // this rank's part of the piece [off, off + n)
        const size_t lo = std::min(std::max(own_lo, off), off + n) - off, hi = std::max(std::min(own_hi, off + n), off) - off;
        if (!bcast || cm->rank == root)
            PMF_HIP_CHECK(hipMemcpyAsync(cm->stage(cm->rank), (const char *)send + off, n, hipMemcpyDeviceToHost, cm->stream));
        ShmOp *o = new (std::nothrow) ShmOp{cm, n, lo, hi > lo ? hi : lo, dtype, op, root};
        PMF_REQUIRE(o, PMF_ENOMEM, "hostshm transport: out of host memory");
        hipError_t e = hipLaunchHostFunc(cm->stream, shm_host_step, o);
        if (e != hipSuccess) {
            delete o;
            pmf_set_error("hipLaunchHostFunc failed: %s", hipGetErrorString(e));
            return PMF_EHIP;
        }
        if (bcast)
            PMF_HIP_CHECK(hipMemcpyAsync((char *)recv + off, cm->h_result, n, hipMemcpyHostToDevice, cm->stream));
        else if (hi > lo)
            PMF_HIP_CHECK(hipMemcpyAsync((char *)recv + off + lo, (char *)cm->h_result + lo, hi - lo, hipMemcpyHostToDevice, cm->stream));
    }
    return PMF_OK;
}

int shm_open_region(PmfComm *cm, const void *unique_id) {
    const unsigned char *id = (const unsigned char *)unique_id;
    unsigned long long h = 1469598103934665603ull;   // FNV-1a of the id: one name per communicator
    for (int k = 0; k < PMF_UNIQUE_ID_BYTES; ++k) h = (h ^ id[k]) * 1099511628211ull;
    snprintf(cm->shm_name, sizeof(cm->shm_name), "/pmf_hip_%016llx", h);
    cm->shm_bytes = kShmHeader + (size_t)cm->nranks * kShmStage;
    int fd = -1;
    if (cm->rank == 0) {
        (void)shm_unlink(cm->shm_name);
        fd = shm_open(cm->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        PMF_REQUIRE(fd >= 0, PMF_ECOMM, "shm_open(%s) failed: %s", cm->shm_name, strerror(errno));
        if (ftruncate(fd, (off_t)cm->shm_bytes) != 0) {
            pmf_set_error("ftruncate(%s) failed: %s", cm->shm_name, strerror(errno));
            close(fd);
            shm_unlink(cm->shm_name);
            return PMF_ECOMM;
        }
    } else {
        const double t0 = now_s();
        for (;;) {   // the creator may not have sized the region yet
            fd = shm_open(cm->shm_name, O_RDWR, 0600);
            if (fd >= 0) {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size == cm->shm_bytes) break;
                close(fd);
                fd = -1;
            }
            PMF_REQUIRE(now_s() - t0 < cm->shm_deadline(), PMF_ECOMM, "hostshm transport: rank 0 never created %s", cm->shm_name);
            usleep(2000);
        }
    }
    // from here on every error exit of rank 0 removes the name: a failed launch must not leave
    // 4 KB + nranks x 32 MB behind in /dev/shm
    auto fail = [&](int code) {
        if (cm->rank == 0) (void)shm_unlink(cm->shm_name);
        return code;
    };
    cm->shm = mmap(nullptr, cm->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (cm->shm == MAP_FAILED) {
        cm->shm = nullptr;
        pmf_set_error("mmap(%s) failed: %s", cm->shm_name, strerror(errno));
        return fail(PMF_ECOMM);
    }
    // pinned staging makes the D2H / H2D legs truly asynchronous; pageable still works
    cm->shm_registered = hipHostRegister(cm->shm, cm->shm_bytes, hipHostRegisterDefault) == hipSuccess;
    if (!cm->shm_registered) (void)hipGetLastError();
    hipError_t he = hipHostMalloc(&cm->h_result, kShmStage, hipHostMallocDefault);
    if (he != hipSuccess) {
        pmf_set_error("hipHostMalloc failed: %s", hipGetErrorString(he));
        return fail(PMF_EHIP);
    }
    // rendezvous: the name can go once every rank has mapped the region
    cm->hdr()->attached.fetch_add(1);
    const double t0 = now_s();
    while (cm->hdr()->attached.load() < cm->nranks) {
        if (now_s() - t0 >= cm->shm_deadline()) {
            pmf_set_error("hostshm transport: %d of %d ranks attached", cm->hdr()->attached.load(), cm->nranks);
            cm->hdr()->failed.store(1);
            return fail(PMF_ECOMM);
        }
        usleep(1000);
    }
    if (cm->rank == 0) (void)shm_unlink(cm->shm_name);
    return PMF_OK;
}

}  // namespace
#endif  // PMF_TEST_TRANSPORT

namespace {

// ---------------------------------------------------------------------------
// transport-independent primitives (all asynchronous on cm->stream)
// ---------------------------------------------------------------------------
// Wait for `stream` while watching the communicator: RCCL reports a dead peer / a failed link through
// ncclCommGetAsyncError, and a rank that never reaches its collective shows up only as time passing.
int comm_wait(PmfComm *cm, hipStream_t stream, const char *what) {
    const double t0 = now_s();
    for (unsigned spins = 0;; ++spins) {
        hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            pmf_set_error("%s: hipStreamQuery failed: %s", what, hipGetErrorString(q));
            return PMF_EHIP;
        }
        if (spins < 2000) continue;            // short waits (the small host collectives) stay on the fast path
        if (cm->nccl) {
            ncclResult_t async = ncclSuccess;
            if (ncclCommGetAsyncError(cm->nccl, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                cm->failed = true;
                pmf_set_error("%s: the communicator reported %s (rank %d of %d); aborting it", what,
                              ncclGetErrorString(async), cm->rank, cm->nranks);
                (void)ncclCommAbort(cm->nccl);
                cm->nccl = nullptr;
                return PMF_ECOMM;
            }
        }
        if (cm->peer_failed()) {
            cm->failed = true;
            pmf_set_error("%s: hostshm transport: a peer failed or timed out", what);
            return PMF_ECOMM;
        }
        if (cm->timeout_s > 0 && now_s() - t0 > cm->timeout_s) {
            cm->failed = true;
            pmf_set_error("%s: no progress for %.0f s (rank %d of %d; PMF_COMM_TIMEOUT_S) -- a peer died or never reached "
                          "its collective", what, cm->timeout_s, cm->rank, cm->nranks);
            if (cm->nccl) {
                (void)ncclCommAbort(cm->nccl);
                cm->nccl = nullptr;
            }
            return PMF_ECOMM;
        }
        usleep(spins < 20000 ? 20 : 500);
    }
    return PMF_OK;
}

int comm_allreduce(PmfComm *cm, void *buf, size_t count, int dtype, int op) {
    if (count == 0) return PMF_OK;
    PMF_REQUIRE(!cm->failed, PMF_ECOMM, "the communicator has failed earlier (a collective error or timeout)");
#ifdef PMF_TEST_TRANSPORT
    if (cm->transport == PMF_TRANSPORT_HOSTSHM)
        return shm_collective(cm, buf, buf, count * (dtype == PMF_F64 ? 8 : 4), dtype, op, 0);
#endif
    PMF_NCCL_CHECK(ncclAllReduce(buf, buf, count, dtype == PMF_F64 ? ncclFloat64 : ncclFloat32,
                                 op == PMF_OP_MAX ? ncclMax : ncclSum, cm->nccl, cm->stream));
    return PMF_OK;
}

int comm_broadcast(PmfComm *cm, const void *send, void *recv, size_t bytes, int root) {
    if (bytes == 0) return PMF_OK;
    PMF_REQUIRE(!cm->failed, PMF_ECOMM, "the communicator has failed earlier (a collective error or timeout)");
#ifdef PMF_TEST_TRANSPORT
    if (cm->transport == PMF_TRANSPORT_HOSTSHM) return shm_collective(cm, send, recv, bytes, -1, 0, root);
#endif
    PMF_NCCL_CHECK(ncclBroadcast(send, recv, bytes, ncclInt8, root, cm->nccl, cm->stream));
    return PMF_OK;
}

// In-place sum reduce-scatter of nranks x `count` elements at `buf`: rank r ends with the sums of slice r
// (elements [r count, (r + 1) count)); the other slices of its buffer are left undefined.
int comm_reduce_scatter(PmfComm *cm, void *buf, size_t count, int dtype) {
    if (count == 0) return PMF_OK;
    PMF_REQUIRE(!cm->failed, PMF_ECOMM, "the communicator has failed earlier (a collective error or timeout)");
    const size_t esz = dtype == PMF_F64 ? 8 : 4;
#ifdef PMF_TEST_TRANSPORT
    if (cm->transport == PMF_TRANSPORT_HOSTSHM)
        return shm_collective(cm, buf, buf, (size_t)cm->nranks * count * esz, dtype, PMF_OP_SUM, 0,
                              (size_t)cm->rank * count * esz, (size_t)(cm->rank + 1) * count * esz);
#endif
    PMF_NCCL_CHECK(ncclReduceScatter(buf, (char *)buf + (size_t)cm->rank * count * esz, count,
                                     dtype == PMF_F64 ? ncclFloat64 : ncclFloat32, ncclSum, cm->nccl, cm->stream));
    return PMF_OK;
}

// In-place all-gather of nranks slices of `bytes` bytes at `buf` (rank r contributes slice r).
int comm_all_gather(PmfComm *cm, void *buf, size_t bytes) {
    if (bytes == 0) return PMF_OK;
    PMF_REQUIRE(!cm->failed, PMF_ECOMM, "the communicator has failed earlier (a collective error or timeout)");
#ifdef PMF_TEST_TRANSPORT
    if (cm->transport == PMF_TRANSPORT_HOSTSHM) {
        for (int r = 0; r < cm->nranks; ++r) {
            char *at = (char *)buf + (size_t)r * bytes;
            int rc = shm_collective(cm, at, at, bytes, -1, 0, r);
            if (rc) return rc;
        }
        return PMF_OK;
    }
#endif
    PMF_NCCL_CHECK(ncclAllGather((char *)buf + (size_t)cm->rank * bytes, buf, bytes, ncclInt8, cm->nccl, cm->stream));
    return PMF_OK;
}

void comm_free(PmfComm *cm) {
    if (!cm) return;
    (void)hipSetDevice(cm->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->nccl) (void)ncclCommDestroy(cm->nccl);
    for (auto &e : cm->ev_ready) (void)hipEventDestroy(e);
    for (auto &e : cm->ev_done) (void)hipEventDestroy(e);
    for (auto &e : cm->ev_fin) (void)hipEventDestroy(e);
    for (auto &e : cm->ev_gath) (void)hipEventDestroy(e);
    if (cm->fin_stream) {
        (void)hipStreamSynchronize(cm->fin_stream);
        (void)hipStreamDestroy(cm->fin_stream);
    }
    if (cm->d_small) (void)hipFree(cm->d_small);
    if (cm->h_small) (void)hipHostFree(cm->h_small);
#ifdef PMF_TEST_TRANSPORT
    if (cm->h_result) (void)hipHostFree(cm->h_result);
    if (cm->shm) {
        if (cm->shm_registered) (void)hipHostUnregister(cm->shm);
        munmap(cm->shm, cm->shm_bytes);
    }
#endif
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
}

int comm_create(pmf_ctx *ctx, int nranks, int rank, const void *unique_id, int transport) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_init: null context");
    PMF_REQUIRE(unique_id != nullptr, PMF_EINVAL, "pmf_comm_init: null unique id");
    PMF_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, PMF_EINVAL, "pmf_comm_init: rank %d of %d", rank, nranks);
    PMF_REQUIRE(ctx->comm == nullptr, PMF_EINVAL, "pmf_comm_init: the context already has a communicator");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PmfComm *cm = new (std::nothrow) PmfComm();
    PMF_REQUIRE(cm, PMF_ENOMEM, "pmf_comm_init: out of host memory");
    cm->nranks = nranks;
    cm->rank = rank;
    cm->device = ctx->device;
    cm->transport = transport;
    if (const char *e = getenv("PMF_COMM_TIMEOUT_S")) {
        cm->timeout_s = atof(e);
        cm->timeout_from_env = true;
    }
    int rc = PMF_OK;
    do {
        int lo = 0, hi = 0;   // collectives must not queue behind a 60 ms accumulate grid
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipError_t e = hipStreamCreateWithPriority(&cm->stream, hipStreamNonBlocking, hi);
        if (e != hipSuccess) {
            pmf_set_error("hipStreamCreateWithPriority failed: %s", hipGetErrorString(e));
            rc = PMF_EHIP;
            break;
        }
        e = hipStreamCreateWithFlags(&cm->fin_stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            pmf_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
            rc = PMF_EHIP;
            break;
        }
        if (hipMalloc(&cm->d_small, kSmallBytes) != hipSuccess ||
            hipHostMalloc(&cm->h_small, kSmallBytes, hipHostMallocDefault) != hipSuccess) {
            pmf_set_error("pmf_comm_init: cannot allocate the collective staging buffers");
            rc = PMF_ENOMEM;
            break;
        }
        if (transport == PMF_TRANSPORT_HOSTSHM) {
#ifdef PMF_TEST_TRANSPORT
            rc = shm_open_region(cm, unique_id);
#else
            pmf_set_error("pmf_comm_init: this build of the library has no hostshm transport (tests use libpmf_hip_test.so)");
            rc = PMF_EINVAL;
#endif
        } else {
            ncclUniqueId id;
            static_assert(sizeof(id) == PMF_UNIQUE_ID_BYTES, "PMF_UNIQUE_ID_BYTES must match ncclUniqueId");
            memcpy(&id, unique_id, sizeof(id));
            ncclResult_t r = ncclCommInitRank(&cm->nccl, nranks, id, rank);
            if (r != ncclSuccess) {
                cm->nccl = nullptr;
                pmf_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, ncclGetErrorString(r));
                rc = PMF_ECOMM;
            }
        }
    } while (0);
    if (rc) {
        comm_free(cm);
        return rc;
    }
    cm->refs = 1;
    ctx->comm = cm;
    return PMF_OK;
}

int ensure_events(PmfComm *cm, size_t n) {
    while (cm->ev_fin.size() < n) {
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 4; ++k) {
            hipError_t e = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int q = 0; q < k; ++q) (void)hipEventDestroy(ev[q]);
                pmf_set_error("hipEventCreate failed: %s", hipGetErrorString(e));
                return PMF_EHIP;
            }
        }
        cm->ev_ready.push_back(ev[0]);
        cm->ev_done.push_back(ev[1]);
        cm->ev_fin.push_back(ev[2]);
        cm->ev_gath.push_back(ev[3]);
    }
    return PMF_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// internal interface used by the sweep translation units
// ---------------------------------------------------------------------------
// A context with a communicator always takes the three-stage path, also with one rank (the all-reduce is
// then RCCL's identity): callers attach one only for multi-rank runs, and the one-GPU tests can drive
// the real RCCL call sequence.
bool pmf_comm_active(const pmf_ctx *ctx) { return ctx->comm != nullptr; }

int pmf_comm_wait_stream(pmf_ctx *ctx, hipStream_t stream, const char *what) { return comm_wait(ctx->comm, stream, what); }

void pmf_comm_release(pmf_ctx *ctx) {
    for (int k = 0; k < 2; ++k) {
        pmf_dev_free(ctx, ctx->d_stats[k], ctx->stats_bytes[k]);
        ctx->d_stats[k] = nullptr;
        ctx->stats_bytes[k] = 0;
    }
    PmfComm *cm = ctx->comm;
    ctx->comm = nullptr;
    if (cm && --cm->refs == 0) comm_free(cm);
}

int pmf_comm_stats(pmf_ctx *ctx, int which, size_t bytes, void **out) {
    *out = nullptr;
    if (ctx->stats_bytes[which] < bytes) {
        if (ctx->d_stats[which]) {
            PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            PMF_HIP_CHECK(hipStreamSynchronize(ctx->comm->stream));
            PMF_HIP_CHECK(hipStreamSynchronize(ctx->comm->fin_stream));
            pmf_dev_free(ctx, ctx->d_stats[which], ctx->stats_bytes[which]);
            ctx->d_stats[which] = nullptr;
            ctx->stats_bytes[which] = 0;
        }
        int rc = pmf_dev_alloc(ctx, &ctx->d_stats[which], bytes);
        if (rc) return rc;
        ctx->stats_bytes[which] = bytes;
    }
    *out = ctx->d_stats[which];
    return PMF_OK;
}

// accumulate(c) -> exchange(c) -> finalize(c), pipelined over the row chunks of `side` (`chunked` = false: one
// message for all rows).  `width` = statistics elements per row.  Three streams, ordered by events only:
//   compute stream      accumulate(0), accumulate(1), ...                       (HBM-bound gathers)
//   collective stream   the collectives of chunk c as soon as accumulate(c) has finished      (xGMI)
//   finalize stream     finalize(c) as soon as chunk c's statistics have landed  (row solves: VALU / LDS-bound)
// so the collective of chunk c AND its finalisation run beside the accumulation of the later chunks; the compute
// stream only waits at the end, for what it has not already been overtaken by.
//
// ALLREDUCE exchange: ncclAllReduce of the chunk's slice, then every rank finalises every row of it (at BASELINE
// config C4 that is 1M 128 x 128 row solves per rank and iteration).
// SCATTER_GATHER exchange (`ex.arrays` = the state arrays finalize writes): the chunk's rows are cut into nranks equal
// sub-ranges of `per` rows (the < nranks rows left over are all-reduced and finalised by everybody);
// ncclReduceScatter leaves rank r with the sums of sub-range r, rank r finalises those rows only, and one
// ncclAllGather per state array hands the finalised rows to everybody -- each rank solves 1/N of the rows and all
// ranks hold the same finalised bytes.  Collective stream order (the same on every rank): RS(0), RS(1), AG(0),
// RS(2), AG(1), ... so the reduce-scatter of chunk c + 1 never queues behind the finalisation of chunk c.
int pmf_comm_half_sweep(pmf_ctx *ctx, int side, size_t width, void *stats, bool chunked,
                        const std::function<int()> &accumulate, const std::function<int()> &finalize,
                        const PmfExchange &ex) {
    PmfComm *cm = ctx->comm;
    const int n = chunked ? ctx->n_chunks[side] : 1;
    int rc = ensure_events(cm, (size_t)n);
    if (rc) return rc;
    const bool sg = ex.n_arrays > 0 && (ctx->exchange == PMF_EXCHANGE_SCATTER_GATHER ||
                                        (ctx->exchange == PMF_EXCHANGE_AUTO && ex.prefer_scatter && cm->nranks > 1));
    const int64_t N = cm->nranks;
    const int saved = ctx->cur_chunk[side];
    hipStream_t const compute = ctx->stream;
    auto fail = [&](const char *what, hipError_t e) {
        pmf_set_error("%s failed: %s", what, hipGetErrorString(e));
        return PMF_EHIP;
    };
    auto chunk_rows = [&](int c, int64_t &r0, int64_t &r1) {
        r0 = chunked ? pmf_chunk_row0(ctx, side, c) : 0;
        r1 = chunked ? pmf_chunk_row0(ctx, side, c + 1) : ctx->rows[side];
    };
    // finalize rows [a, b) of the selected chunk on the finalize stream
    auto finalize_rows = [&](int64_t a, int64_t b) {
        if (b <= a) return (int)PMF_OK;
        ctx->fin_row0 = a;
        ctx->fin_row1 = b;
        ctx->stream = cm->fin_stream;
        const int r = finalize();
        ctx->stream = compute;
        ctx->fin_row0 = ctx->fin_row1 = -1;
        return r;
    };
    // all-gather of chunk c's finalised state (SCATTER_GATHER), behind its finalisation
    auto gather = [&](int c) {
        int64_t r0, r1;
        chunk_rows(c, r0, r1);
        const int64_t per = (r1 - r0) / N;
        hipError_t e = hipStreamWaitEvent(cm->stream, cm->ev_fin[(size_t)c], 0);
        if (e != hipSuccess) return fail("event ordering of the state all-gather", e);
        int r = PMF_OK;
        pmf_prof_begin_on(ctx, PMF_KERNEL_COMM_ALLREDUCE, cm->stream);
        for (int k = 0; k < ex.n_arrays && !r && per > 0; ++k) {
            int host_width, stride;
            pmf_array_shape(ctx, ex.arrays[k], &host_width, &stride);
            const size_t row_bytes = (size_t)stride * ctx->elem;
            r = comm_all_gather(cm, (char *)ctx->arr[side][ex.arrays[k]] + (size_t)r0 * row_bytes, (size_t)per * row_bytes);
        }
        pmf_prof_end_on(ctx, cm->stream);
        if (r) return r;
        e = hipEventRecord(cm->ev_gath[(size_t)c], cm->stream);
        return e == hipSuccess ? (int)PMF_OK : fail("hipEventRecord", e);
    };
    int issued = 0;     // chunks whose finalize has been queued
    int gathered = 0;   // chunks whose all-gather has been queued
    for (int c = 0; c < n && !rc; ++c) {
        ctx->cur_chunk[side] = chunked ? c : -1;
        if ((rc = accumulate())) break;
        int64_t r0, r1;
        chunk_rows(c, r0, r1);
        hipError_t e = hipEventRecord(cm->ev_ready[(size_t)c], compute);
        if (e == hipSuccess) e = hipStreamWaitEvent(cm->stream, cm->ev_ready[(size_t)c], 0);
        if (e != hipSuccess) {
            rc = fail("event ordering of the item collective", e);
            break;
        }
        const int64_t per = sg ? (r1 - r0) / N : 0, main = per * N;   // rows [r0, r0 + main) are scattered
        pmf_prof_begin_on(ctx, PMF_KERNEL_COMM_ALLREDUCE, cm->stream);
        if (per > 0) rc = comm_reduce_scatter(cm, (char *)stats + (size_t)r0 * width * ctx->elem, (size_t)per * width, ctx->dtype);
        if (!rc)
            rc = comm_allreduce(cm, (char *)stats + (size_t)(r0 + main) * width * ctx->elem, (size_t)(r1 - r0 - main) * width,
                                ctx->dtype, PMF_OP_SUM);
        pmf_prof_end_on(ctx, cm->stream);
        if (rc) break;
        e = hipEventRecord(cm->ev_done[(size_t)c], cm->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cm->fin_stream, cm->ev_done[(size_t)c], 0);
        if (e != hipSuccess) {
            rc = fail("event ordering of the finalize", e);
            break;
        }
        // the finalize kernels of this chunk go to the finalize stream: this rank's sub-range, then the rows all ranks share
        if (per > 0) rc = finalize_rows(r0 + cm->rank * per, r0 + (cm->rank + 1) * per);
        if (!rc) rc = finalize_rows(r0 + main, r1);
        if (rc) break;
        e = hipEventRecord(cm->ev_fin[(size_t)c], cm->fin_stream);
        if (e != hipSuccess) {
            rc = fail("hipEventRecord", e);
            break;
        }
        issued = c + 1;
        if (sg && c > 0) {   // the previous chunk's gather goes behind this chunk's reduce-scatter
            if ((rc = gather(c - 1))) break;
            gathered = c;
        }
    }
    if (sg && !rc && issued == n) {
        rc = gather(n - 1);
        if (!rc) gathered = n;
    }
    // the next half-sweep reads the finalised rows: the compute stream joins the finalize (and gather) work here.
    // Its idle time in these waits is the communication + finalisation that accumulation did not hide.
    for (int c = 0; c < issued; ++c) {
        pmf_prof_begin_on(ctx, PMF_KERNEL_COMM_WAIT, compute);
        hipError_t e = hipStreamWaitEvent(compute, cm->ev_fin[(size_t)c], 0);
        if (e == hipSuccess && c < gathered) e = hipStreamWaitEvent(compute, cm->ev_gath[(size_t)c], 0);
        pmf_prof_end_on(ctx, compute);
        if (e != hipSuccess && !rc) rc = fail("hipStreamWaitEvent", e);
    }
    ctx->stream = compute;
    ctx->cur_chunk[side] = saved;
    ctx->fin_row0 = ctx->fin_row1 = -1;
    if (rc) {   // leave nothing half-ordered behind an error
        (void)hipStreamSynchronize(cm->stream);
        (void)hipStreamSynchronize(cm->fin_stream);
        (void)hipStreamSynchronize(compute);
    }
    return rc;
}

// ---------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------
extern "C" int pmf_comm_unique_id(void *id_out) {
    PMF_REQUIRE(id_out != nullptr, PMF_EINVAL, "pmf_comm_unique_id: null argument");
    ncclUniqueId id;
    PMF_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return PMF_OK;
}

extern "C" int pmf_comm_init(pmf_ctx *ctx, int nranks, int rank, const void *unique_id) {
    return comm_create(ctx, nranks, rank, unique_id, PMF_TRANSPORT_RCCL);
}

#ifdef PMF_TEST_TRANSPORT
extern "C" int pmf_comm_init_hostshm(pmf_ctx *ctx, int nranks, int rank, const void *unique_id) {
    return comm_create(ctx, nranks, rank, unique_id, PMF_TRANSPORT_HOSTSHM);
}
#endif

extern "C" int pmf_comm_set_exchange(pmf_ctx *ctx, int mode) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_set_exchange: null context");
    PMF_REQUIRE(mode == PMF_EXCHANGE_AUTO || mode == PMF_EXCHANGE_ALLREDUCE || mode == PMF_EXCHANGE_SCATTER_GATHER, PMF_EINVAL,
                "pmf_comm_set_exchange: bad mode %d", mode);
    ctx->exchange = mode;
    return PMF_OK;
}

extern "C" int pmf_comm_attach(pmf_ctx *ctx, pmf_ctx *owner) {
    PMF_REQUIRE(ctx != nullptr && owner != nullptr, PMF_EINVAL, "pmf_comm_attach: null context");
    PMF_REQUIRE(owner->comm != nullptr, PMF_EINVAL, "pmf_comm_attach: the owner has no communicator");
    PMF_REQUIRE(ctx->comm == nullptr, PMF_EINVAL, "pmf_comm_attach: the context already has a communicator");
    PMF_REQUIRE(ctx->device == owner->device, PMF_EINVAL, "pmf_comm_attach: contexts on different devices (%d, %d)",
                ctx->device, owner->device);
    ctx->comm = owner->comm;
    ctx->comm->refs += 1;
    return PMF_OK;
}

extern "C" int pmf_comm_destroy(pmf_ctx *ctx) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_destroy: null context");
    if (!ctx->comm) return PMF_OK;
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    pmf_comm_release(ctx);
    return PMF_OK;
}

extern "C" int pmf_comm_info(pmf_ctx *ctx, int *nranks, int *rank, int *transport) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_info: null context");
    if (nranks) *nranks = ctx->comm ? ctx->comm->nranks : 1;
    if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
    if (transport) *transport = ctx->comm ? ctx->comm->transport : -1;
    return PMF_OK;
}

// Element-wise reduction of `n` host doubles over the ranks (sum or max), result on every rank.
// Orders nothing against the context's compute stream: callers reduce values they already hold.
extern "C" int pmf_comm_allreduce_host(pmf_ctx *ctx, double *values, int64_t n, int op) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_allreduce_host: null context");
    PMF_REQUIRE(ctx->comm != nullptr, PMF_EINVAL, "pmf_comm_allreduce_host: the context has no communicator");
    PMF_REQUIRE(n >= 0 && (values != nullptr || n == 0), PMF_EINVAL, "pmf_comm_allreduce_host: bad arguments");
    PMF_REQUIRE(op == PMF_OP_SUM || op == PMF_OP_MAX, PMF_EINVAL, "pmf_comm_allreduce_host: bad op %d", op);
    PmfComm *cm = ctx->comm;
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t step = (int64_t)(kSmallBytes / sizeof(double));
    for (int64_t at = 0; at < n; at += step) {
        const int64_t m = std::min(step, n - at);
        memcpy(cm->h_small, values + at, (size_t)m * sizeof(double));
        PMF_HIP_CHECK(hipMemcpyAsync(cm->d_small, cm->h_small, (size_t)m * sizeof(double), hipMemcpyHostToDevice, cm->stream));
        int rc = comm_allreduce(cm, cm->d_small, (size_t)m, PMF_F64, op);
        if (rc) return rc;
        PMF_HIP_CHECK(hipMemcpyAsync(cm->h_small, cm->d_small, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, cm->stream));
        if ((rc = comm_wait(cm, cm->stream, "pmf_comm_allreduce_host"))) return rc;
        PMF_REQUIRE(!cm->peer_failed(), PMF_ECOMM, "hostshm transport: a peer failed or timed out");
        memcpy(values + at, cm->h_small, (size_t)m * sizeof(double));
    }
    return PMF_OK;
}

// Every rank's queued work (compute and collectives) has finished when this returns on any rank.
extern "C" int pmf_comm_barrier(pmf_ctx *ctx) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_barrier: null context");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->comm) {   // the streams that can be stuck behind a collective are waited for under the watchdog first
        int rc;
        if ((rc = comm_wait(ctx->comm, ctx->stream, "pmf_comm_barrier"))) return rc;
        if ((rc = comm_wait(ctx->comm, ctx->comm->stream, "pmf_comm_barrier"))) return rc;
        if ((rc = comm_wait(ctx->comm, ctx->comm->fin_stream, "pmf_comm_barrier"))) return rc;
    }
    PMF_HIP_CHECK(hipDeviceSynchronize());   // every stream of this process on the device, other contexts' included
    if (!ctx->comm) return PMF_OK;
    double one = 1.0;
    return pmf_comm_allreduce_host(ctx, &one, 1, PMF_OP_SUM);
}

// The user-side rows of every rank, in rank order, as host float64 on every rank (after a sharded fit
// each rank holds one user range).  bounds[r] .. bounds[r + 1] = rank r's global user range.
extern "C" int pmf_comm_gather_user_rows(pmf_ctx *ctx, int array, const int64_t *bounds, double *host_full) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_comm_gather_user_rows: null context");
    PMF_REQUIRE(ctx->comm != nullptr, PMF_EINVAL, "pmf_comm_gather_user_rows: the context has no communicator");
    PMF_REQUIRE(array >= 0 && array < PMF_ARR_COUNT, PMF_EINVAL, "pmf_comm_gather_user_rows: bad array id %d", array);
    PMF_REQUIRE(bounds != nullptr && host_full != nullptr, PMF_EINVAL, "pmf_comm_gather_user_rows: null argument");
    PmfComm *cm = ctx->comm;
    PMF_REQUIRE(bounds[cm->rank + 1] - bounds[cm->rank] == ctx->rows[PMF_SIDE_USER], PMF_EINVAL,
                "pmf_comm_gather_user_rows: this rank's range [%lld, %lld) does not match its %lld users",
                (long long)bounds[cm->rank], (long long)bounds[cm->rank + 1], (long long)ctx->rows[PMF_SIDE_USER]);
    for (int r = 0; r < cm->nranks; ++r)
        PMF_REQUIRE(bounds[r + 1] >= bounds[r], PMF_EINVAL, "pmf_comm_gather_user_rows: bounds must not decrease");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = pmf_require_array(ctx, PMF_SIDE_USER, array, "pmf_comm_gather_user_rows");
    if (rc) return rc;
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int width, stride;
    pmf_array_shape(ctx, array, &width, &stride);
    const int64_t row_bytes = (int64_t)stride * (int64_t)ctx->elem;
    const int64_t step = std::max<int64_t>(1, (64ll << 20) / row_bytes);
    int64_t longest = 0;
    for (int r = 0; r < cm->nranks; ++r) longest = std::max(longest, bounds[r + 1] - bounds[r]);
    const size_t block = (size_t)(std::min(step, std::max<int64_t>(longest, 1)) * row_bytes);
    if ((rc = pmf_ensure_scratch(ctx, block))) return rc;
    if ((rc = pmf_ensure_pinned(ctx, block))) return rc;
    for (int r = 0; r < cm->nranks; ++r) {
        const int64_t rows = bounds[r + 1] - bounds[r];
        for (int64_t r0 = 0; r0 < rows; r0 += step) {
            const int64_t nr = std::min(step, rows - r0);
            const size_t bytes = (size_t)(nr * row_bytes);
            const char *src = r == cm->rank ? (const char *)ctx->arr[PMF_SIDE_USER][array] + r0 * row_bytes
                                            : (const char *)ctx->d_scratch;
            if ((rc = comm_broadcast(cm, src, ctx->d_scratch, bytes, r))) return rc;
            PMF_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, bytes, hipMemcpyDeviceToHost, cm->stream));
            if ((rc = comm_wait(cm, cm->stream, "pmf_comm_gather_user_rows"))) return rc;
            pmf_unpack_rows(ctx, array, ctx->h_pinned, host_full + (bounds[r] + r0) * width, nr);
        }
    }
    PMF_REQUIRE(!cm->peer_failed(), PMF_ECOMM, "hostshm transport: a peer failed or timed out");
    return PMF_OK;
}
