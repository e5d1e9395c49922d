// Multi-GPU side of libpmf_hip.so (SURVEY.md section 8(b) `pmf_comm_init`, section 8(e)).
//
// Ratings are sharded by user range, one process per GPU; the item block is replicated.  An item
// half-sweep is  accumulate (raw per-item sums over this rank's ratings) -> all-reduce -> finalize.
// This file owns everything around the collective: the RCCL communicator, a second (high-priority)
// HIP stream the collectives run on, the event ordering between that stream and the context's
// compute stream, the library-owned statistics buffers, and the row-chunk pipeline that lets the
// all-reduce of chunk c travel over xGMI while chunk c+1 is being accumulated.
//
// Two transports sit behind one small interface:
//   RCCL     ncclAllReduce / ncclBroadcast on the collective stream (one rank per GPU; production);
//   HOSTSHM  ranks of ONE GPU exchange through POSIX shared memory in stream order
//            (hipLaunchHostFunc).  RCCL refuses two ranks on one device, so this is how the multi-rank
//            code path -- chunk offsets, event ordering, the sharded fit -- is rehearsed on a
//            one-GPU box.  It is a test transport: slow, node-local, never chosen by default.
#include <errno.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

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

constexpr size_t kShmHeader = 4096;
constexpr size_t kShmStage = 32u << 20;    // staging bytes per rank (messages travel in pieces of this size)
constexpr double kShmTimeoutS = 120.0;     // a peer that never arrives must not hang the box
constexpr size_t kSmallBytes = 64u << 10;  // device / pinned buffer of the host-value collectives

struct ShmHeader {
    std::atomic<int> count;
    std::atomic<int> generation;
    std::atomic<int> attached;
    std::atomic<int> failed;
};

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
    std::vector<hipEvent_t> ev_ready, ev_done, ev_fin;  // per row chunk: statistics ready / all-reduced / finalized
    void *d_small = nullptr;                  // kSmallBytes, host-value collectives
    void *h_small = nullptr;                  // pinned twin
    bool failed = false;                      // a collective failed or timed out: every later call returns PMF_ECOMM
    double timeout_s = 1800.0;                // PMF_COMM_TIMEOUT_S
    // HOSTSHM
    char shm_name[64] = "";
    void *shm = nullptr;
    size_t shm_bytes = 0;
    bool shm_registered = false;
    void *h_result = nullptr;                 // pinned, kShmStage: this rank's reduced piece
    ShmHeader *hdr() const { return (ShmHeader *)shm; }
    char *stage(int r) const { return (char *)shm + kShmHeader + (size_t)r * kShmStage; }
};

// ---------------------------------------------------------------------------
// HOSTSHM transport
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
            if (now_s() - t0 > (cm->timeout_s > 0 ? std::min(kShmTimeoutS, cm->timeout_s) : kShmTimeoutS)) {
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
    size_t bytes;
    int dtype;   // PMF_F32 / PMF_F64; -1 = broadcast
    int op;      // PMF_OP_SUM / PMF_OP_MAX
    int root;
};

template <typename T>
void shm_reduce(PmfComm *cm, size_t n, int op) {
    T *out = (T *)cm->h_result;
    const T *first = (const T *)cm->stage(0);
    for (size_t k = 0; k < n; ++k) out[k] = first[k];
    for (int r = 1; r < cm->nranks; ++r) {   // rank order: every rank gets bit-identical sums
        const T *src = (const T *)cm->stage(r);
        if (op == PMF_OP_MAX)
            for (size_t k = 0; k < n; ++k) out[k] = src[k] > out[k] ? src[k] : out[k];
        else
            for (size_t k = 0; k < n; ++k) out[k] += src[k];
    }
}

// runs on the collective stream's callback thread, after this rank's D2H copy of the piece
void shm_host_step(void *arg) {
    ShmOp *o = (ShmOp *)arg;
    PmfComm *cm = o->cm;
    if (shm_barrier(cm)) {                     // every rank's piece is staged
        if (o->dtype < 0) memcpy(cm->h_result, cm->stage(o->root), o->bytes);
        else if (o->dtype == PMF_F64) shm_reduce<double>(cm, o->bytes / 8, o->op);
        else shm_reduce<float>(cm, o->bytes / 4, o->op);
        (void)shm_barrier(cm);                 // nobody overwrites a staging area that is still being read
    }
    delete o;
}

int shm_collective(PmfComm *cm, const void *send, void *recv, size_t bytes, int dtype, int op, int root) {
    PMF_REQUIRE(!cm->hdr()->failed.load(), PMF_ECOMM, "hostshm transport: a peer failed or timed out");
    const bool bcast = dtype < 0;
    for (size_t off = 0; off < bytes; off += kShmStage) {
        const size_t n = std::min(kShmStage, bytes - off);
        if (!bcast || cm->rank == root)
            PMF_HIP_CHECK(hipMemcpyAsync(cm->stage(cm->rank), (const char *)send + off, n, hipMemcpyDeviceToHost, cm->stream));
        ShmOp *o = new (std::nothrow) ShmOp{cm, n, dtype, op, root};
        PMF_REQUIRE(o, PMF_ENOMEM, "hostshm transport: out of host memory");
        hipError_t e = hipLaunchHostFunc(cm->stream, shm_host_step, o);
        if (e != hipSuccess) {
            delete o;
            pmf_set_error("hipLaunchHostFunc failed: %s", hipGetErrorString(e));
            return PMF_EHIP;
        }
        PMF_HIP_CHECK(hipMemcpyAsync((char *)recv + off, cm->h_result, n, hipMemcpyHostToDevice, cm->stream));
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
            PMF_REQUIRE(now_s() - t0 < kShmTimeoutS, PMF_ECOMM, "hostshm transport: rank 0 never created %s", cm->shm_name);
            usleep(2000);
        }
    }
    cm->shm = mmap(nullptr, cm->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (cm->shm == MAP_FAILED) {
        cm->shm = nullptr;
        pmf_set_error("mmap(%s) failed: %s", cm->shm_name, strerror(errno));
        return PMF_ECOMM;
    }
    // pinned staging makes the D2H / H2D legs truly asynchronous; pageable still works
    cm->shm_registered = hipHostRegister(cm->shm, cm->shm_bytes, hipHostRegisterDefault) == hipSuccess;
    if (!cm->shm_registered) (void)hipGetLastError();
    PMF_HIP_CHECK(hipHostMalloc(&cm->h_result, kShmStage, hipHostMallocDefault));
    // rendezvous: the name can go once every rank has mapped the region
    cm->hdr()->attached.fetch_add(1);
    const double t0 = now_s();
    while (cm->hdr()->attached.load() < cm->nranks) {
        PMF_REQUIRE(now_s() - t0 < kShmTimeoutS, PMF_ECOMM, "hostshm transport: %d of %d ranks attached",
                    cm->hdr()->attached.load(), cm->nranks);
        usleep(1000);
    }
    if (cm->rank == 0) (void)shm_unlink(cm->shm_name);
    return PMF_OK;
}

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
        if (cm->transport == PMF_TRANSPORT_HOSTSHM && cm->hdr()->failed.load()) {
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
    if (cm->transport == PMF_TRANSPORT_HOSTSHM)
        return shm_collective(cm, buf, buf, count * (dtype == PMF_F64 ? 8 : 4), dtype, op, 0);
    PMF_NCCL_CHECK(ncclAllReduce(buf, buf, count, dtype == PMF_F64 ? ncclFloat64 : ncclFloat32,
                                 op == PMF_OP_MAX ? ncclMax : ncclSum, cm->nccl, cm->stream));
    return PMF_OK;
}

int comm_broadcast(PmfComm *cm, const void *send, void *recv, size_t bytes, int root) {
    if (bytes == 0) return PMF_OK;
    PMF_REQUIRE(!cm->failed, PMF_ECOMM, "the communicator has failed earlier (a collective error or timeout)");
    if (cm->transport == PMF_TRANSPORT_HOSTSHM) return shm_collective(cm, send, recv, bytes, -1, 0, root);
    PMF_NCCL_CHECK(ncclBroadcast(send, recv, bytes, ncclInt8, root, cm->nccl, cm->stream));
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
    if (cm->fin_stream) {
        (void)hipStreamSynchronize(cm->fin_stream);
        (void)hipStreamDestroy(cm->fin_stream);
    }
    if (cm->d_small) (void)hipFree(cm->d_small);
    if (cm->h_small) (void)hipHostFree(cm->h_small);
    if (cm->h_result) (void)hipHostFree(cm->h_result);
    if (cm->shm) {
        if (cm->shm_registered) (void)hipHostUnregister(cm->shm);
        munmap(cm->shm, cm->shm_bytes);
    }
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
    if (const char *e = getenv("PMF_COMM_TIMEOUT_S")) cm->timeout_s = atof(e);
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
            rc = shm_open_region(cm, unique_id);
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
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < 3; ++k) {
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
        PMF_REQUIRE(!ctx->capturing, PMF_EINVAL, "a statistics buffer would have to grow inside a graph capture");
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

// accumulate(c) -> all-reduce(c) -> finalize(c), pipelined over the row chunks of `side` (`chunked` = false: one
// message for all rows).  `width` = statistics elements per row.  Three streams, ordered by events only:
//   compute stream      accumulate(0), accumulate(1), ...                       (HBM-bound gathers)
//   collective stream   all-reduce(c) as soon as accumulate(c) has finished      (xGMI)
//   finalize stream     finalize(c) as soon as all-reduce(c) has landed          (row solves: VALU / LDS-bound)
// so the collective of chunk c AND its finalisation run beside the accumulation of the later chunks; the compute
// stream only waits at the end, for the finalisations it has not already been overtaken by.  (Every rank
// finalises every item: at BASELINE config C4 that is 1M 128 x 128 row solves per rank and iteration, about
// 0.1 s -- hidden here instead of queued behind the last accumulate.)
int pmf_comm_half_sweep(pmf_ctx *ctx, int side, size_t width, void *stats, bool chunked,
                        const std::function<int()> &accumulate, const std::function<int()> &finalize) {
    PmfComm *cm = ctx->comm;
    PMF_REQUIRE(!ctx->capturing, PMF_EINVAL, "a multi-GPU half-sweep cannot be captured into a HIP graph");
    const int n = chunked ? ctx->n_chunks[side] : 1;
    int rc = ensure_events(cm, (size_t)n);
    if (rc) return rc;
    const int saved = ctx->cur_chunk[side];
    hipStream_t const compute = ctx->stream;
    auto fail = [&](const char *what, hipError_t e) {
        pmf_set_error("%s failed: %s", what, hipGetErrorString(e));
        return PMF_EHIP;
    };
    int issued = 0;   // chunks whose finalize has been queued
    for (int c = 0; c < n && !rc; ++c) {
        ctx->cur_chunk[side] = chunked ? c : -1;
        if ((rc = accumulate())) break;
        const int64_t r0 = chunked ? pmf_chunk_row0(ctx, side, c) : 0;
        const int64_t r1 = chunked ? pmf_chunk_row0(ctx, side, c + 1) : ctx->rows[side];
        hipError_t e = hipEventRecord(cm->ev_ready[(size_t)c], compute);
        if (e == hipSuccess) e = hipStreamWaitEvent(cm->stream, cm->ev_ready[(size_t)c], 0);
        if (e != hipSuccess) {
            rc = fail("event ordering of the item all-reduce", e);
            break;
        }
        pmf_prof_begin_on(ctx, PMF_KERNEL_COMM_ALLREDUCE, cm->stream);
        rc = comm_allreduce(cm, (char *)stats + (size_t)r0 * width * ctx->elem, (size_t)(r1 - r0) * width, ctx->dtype,
                            PMF_OP_SUM);
        pmf_prof_end_on(ctx, cm->stream);
        if (rc) break;
        e = hipEventRecord(cm->ev_done[(size_t)c], cm->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cm->fin_stream, cm->ev_done[(size_t)c], 0);
        if (e != hipSuccess) {
            rc = fail("event ordering of the finalize", e);
            break;
        }
        ctx->stream = cm->fin_stream;     // the finalize kernels of this chunk go to the finalize stream
        rc = finalize();
        ctx->stream = compute;
        if (rc) break;
        e = hipEventRecord(cm->ev_fin[(size_t)c], cm->fin_stream);
        if (e != hipSuccess) {
            rc = fail("hipEventRecord", e);
            break;
        }
        issued = c + 1;
    }
    // the next half-sweep reads the finalised rows: the compute stream joins the finalize stream here.  Its idle
    // time in these waits is the communication + finalisation that accumulation did not hide.
    for (int c = 0; c < issued; ++c) {
        pmf_prof_begin_on(ctx, PMF_KERNEL_COMM_WAIT, compute);
        hipError_t e = hipStreamWaitEvent(compute, cm->ev_fin[(size_t)c], 0);
        pmf_prof_end_on(ctx, compute);
        if (e != hipSuccess && !rc) rc = fail("hipStreamWaitEvent", e);
    }
    ctx->stream = compute;
    ctx->cur_chunk[side] = saved;
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

extern "C" int pmf_comm_init_hostshm(pmf_ctx *ctx, int nranks, int rank, const void *unique_id) {
    return comm_create(ctx, nranks, rank, unique_id, PMF_TRANSPORT_HOSTSHM);
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
        PMF_REQUIRE(cm->transport != PMF_TRANSPORT_HOSTSHM || !cm->hdr()->failed.load(), PMF_ECOMM,
                    "hostshm transport: a peer failed or timed out");
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
    PMF_REQUIRE(cm->transport != PMF_TRANSPORT_HOSTSHM || !cm->hdr()->failed.load(), PMF_ECOMM,
                "hostshm transport: a peer failed or timed out");
    return PMF_OK;
}
