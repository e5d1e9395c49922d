// Internal declarations shared by the translation units of libpmf_hip.so.
// gfx950 only: wave64, 16-byte vector accesses, DPP row reductions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "pmf_hip.h"

#ifndef PMF_TRANSPORT_HOSTSHM
#define PMF_TRANSPORT_HOSTSHM 1   // declared by the header in test builds only (-DPMF_TEST_TRANSPORT)
#endif

#define PMF_WAVE 64
#define PMF_VEC 4               // elements per lane access (16 B fp32 / 32 B fp64)
#define PMF_RATE_FLOOR 1e-10    // hpf_cavi.py:141
#define PMF_MAX_LABELS 32

void pmf_set_error(const char *fmt, ...);

#define PMF_HIP_CHECK(expr)                                                         \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            pmf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                          __FILE__, __LINE__);                                      \
            return PMF_EHIP;                                                        \
        }                                                                           \
    } while (0)

#define PMF_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            pmf_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

// One unit of sweep work: a contiguous run of one row's ratings.
// slot < 0  : the run is the whole row  -> the kernel finalises the row itself
// slot >= 0 : the row is split          -> raw sums go to partial slot `slot`
struct PmfTask {
    int64_t start;   // offset into the side's col/val arrays
    int32_t row;
    int32_t len;
    int32_t slot;
    int32_t pad;
};

// A row whose ratings were split over several tasks (heavy rows).
struct PmfSplitRow {
    int32_t row;
    int32_t first_slot;
    int32_t n_slots;
    int32_t pad;
};

struct PmfTaskList {
    int64_t n_tasks = 0;
    int64_t n_slots = 0;
    int64_t n_split = 0;
    int32_t max_len = 0;
    PmfTask *d_tasks = nullptr;
    PmfSplitRow *d_split = nullptr;
    int32_t *d_split_rows = nullptr;  // row id of every split row (solve list)
    // row chunks (pmf_ctx_set_row_chunks): tasks are grouped by the chunk of their row,
    // longest-first inside each group; [n_chunks + 1] offsets into d_tasks / d_split
    std::vector<int64_t> task_off, split_off;
};

// The part of a task list (and of the row range) one accumulate / finalize call covers.
struct PmfTaskView {
    const PmfTask *d_tasks = nullptr;
    const PmfSplitRow *d_split = nullptr;
    const int32_t *d_split_rows = nullptr;
    int64_t n_tasks = 0, n_split = 0, n_slots = 0;
    int64_t row0 = 0, row1 = 0;            // row range [row0, row1)
    const int32_t *d_nonempty = nullptr;   // rows of the range with at least one rating
    int64_t n_nonempty = 0;
};

// Ratings ordered by one side (CSR when side = user, CSC when side = item).
struct PmfSideIndex {
    int64_t *d_ptr = nullptr;    // [rows + 1]
    int32_t *d_other = nullptr;  // [nnz] id on the opposite side
    void *d_val = nullptr;       // [nnz] rating, context dtype
    std::vector<int64_t> h_ptr;  // host copy of ptr (task building)
    int32_t *d_nonempty = nullptr;  // rows with at least one rating (Gaussian solve list)
    int64_t n_nonempty = 0;
    std::vector<int32_t> h_nonempty;    // host copy of d_nonempty
    std::vector<int64_t> nonempty_off;  // [n_chunks + 1] offsets into d_nonempty
    PmfTaskList gamma_tasks;     // chunk = PMF_GAMMA_CHUNK, empty rows included
    PmfTaskList gauss_tasks;     // chunk = PMF_GAUSS_CHUNK, empty rows excluded
    PmfTaskList bias_tasks;      // chunk <= PMF_GAMMA_CHUNK, empty rows excluded
    PmfTaskList sgd_tasks;       // chunk = PMF_SGD_CHUNK exactly (the gradient mode is defined by it), empty rows excluded
};

struct PmfEvalSet {
    int64_t n = 0;
    int n_labels = 0;
    int32_t *d_u = nullptr;
    int32_t *d_i = nullptr;
    double *d_y = nullptr;
    int32_t *d_label = nullptr;
};

struct pmf_ctx {
    int device = 0;
    int dtype = PMF_F32;
    int64_t rows[2] = {0, 0};
    int K = 0;
    int kpad = 0;        // K rounded up to PMF_VEC
    int kp = 0;          // K(K+1)/2
    int cov_stride = 0;  // kp rounded up to PMF_VEC
    int64_t nnz = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    size_t elem = 4;

    int n_chunks[2] = {1, 1};    // row chunks per side (multi-GPU pipelining of a half-sweep)
    int cur_chunk[2] = {-1, -1};  // chunk the accumulate / finalize calls act on; -1 = all rows
    // row window of a finalize-from-statistics call inside pmf_comm_half_sweep (the sub-range of a chunk this rank
    // owns under the SCATTER_GATHER exchange); -1 = the whole selected chunk
    int64_t fin_row0 = -1, fin_row1 = -1;
    int exchange = PMF_EXCHANGE_AUTO;   // pmf_comm_set_exchange

    void *arr[2][PMF_ARR_COUNT] = {};
    PmfSideIndex index[2];
    PmfEvalSet eval;

    // scratch, grown on demand
    void *d_partial = nullptr;
    size_t partial_bytes = 0;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void *h_pinned = nullptr;
    size_t pinned_bytes = 0;

    int64_t device_bytes = 0;

    // diagnostic switches, read from the environment once when the context is created
    bool gauss_generic = false;    // PMF_GAUSS_GENERIC: the generic accumulate kernel instead of the MFMA ones
    bool gauss_unfused = false;    // PMF_GAUSS_UNFUSED: accumulate and solve as separate launches
    bool gauss_lds_solve = false;  // PMF_GAUSS_LDS_SOLVE: the block-per-row LDS solve for 64 < K <= 128
    int topk_max_blocks = 0;       // PMF_TOPK_MAX_BLOCKS=n caps the fused kernel's persistent grid (tests: many tiles per block)
    int topk_stage_buffers = 0;    // PMF_TOPK_STAGE_BUFFERS=1|2 pins the fused kernel's stage buffering (0: by residency)
    bool topk_two_phase = false;   // PMF_TOPK_TWO_PHASE: score matrix in HBM + select instead of the fused kernel

    // multi-GPU (pmf_comm.hip): the communicator (shared between contexts of one process, refcounted)
    // and the library-owned statistics buffers of the item half-sweeps
    // (0: factor / gamma / gradient statistics, 1: Gaussian bias statistics)
    struct PmfComm *comm = nullptr;
    void *d_stats[2] = {nullptr, nullptr};
    size_t stats_bytes[2] = {0, 0};

    bool prof = false;
    struct ProfRec {
        hipEvent_t a, b;
        int kernel;
    };
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[PMF_KERNEL_COUNT] = {};
    int64_t prof_n[PMF_KERNEL_COUNT] = {};
};

#define PMF_GAMMA_CHUNK 512   // (256 until round 2: 512 halves the split-row slots; HPF K=64 at C3: gamma_final 0.16 -> 0.08 ms)
#define PMF_SGD_CHUNK 256     // the gradient mode's piece length is part of its definition (include/pmf_hip.h)
#define PMF_GAUSS_CHUNK 512

// first row of chunk c of a side (c = n_chunks gives the row count)
static inline int64_t pmf_chunk_row0(const pmf_ctx *ctx, int side, int c) {
    return ctx->rows[side] * (int64_t)c / ctx->n_chunks[side];
}
// `select` = honour pmf_ctx_select_chunk (accumulate / finalize); fused sweeps pass false
PmfTaskView pmf_task_view(const pmf_ctx *ctx, int side, const PmfTaskList &tl, bool select);

int pmf_dev_alloc(pmf_ctx *ctx, void **p, size_t bytes);
void pmf_dev_free(pmf_ctx *ctx, void *p, size_t bytes);
int pmf_ensure_partial(pmf_ctx *ctx, size_t bytes);
int pmf_ensure_scratch(pmf_ctx *ctx, size_t bytes);
int pmf_ensure_pinned(pmf_ctx *ctx, size_t bytes);
size_t pmf_array_elems(const pmf_ctx *ctx, int side, int array);  // device elements
int pmf_require_array(pmf_ctx *ctx, int side, int array, const char *what);
int pmf_alloc_array(pmf_ctx *ctx, int side, int array);  // no-op if present (zero-filled)

// device-side index build (pmf_index.hip)
struct PmfIndexBuild;
int pmf_index_device_begin(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids, const int32_t *item_ids,
                           const double *ratings, PmfIndexBuild **out, int64_t *bad_position);
void pmf_index_device_abort(PmfIndexBuild *b);
int pmf_index_device_finish(pmf_ctx *ctx, PmfIndexBuild *b, int64_t nnz);

// host <-> device layout helpers (pmf_ctx.hip)
void pmf_array_shape(const pmf_ctx *ctx, int array, int *host_width, int *dev_stride);
void pmf_unpack_rows(const pmf_ctx *ctx, int array, const void *src, double *dst, int64_t rows);

// multi-GPU (pmf_comm.hip).  With an attached communicator of more than one rank the ITEM half-sweeps
// run  accumulate -> all-reduce -> finalize  through pmf_comm_half_sweep.
bool pmf_comm_active(const pmf_ctx *ctx);
void pmf_comm_release(pmf_ctx *ctx);   // detach + free the statistics buffers (pmf_ctx_destroy)
int pmf_comm_stats(pmf_ctx *ctx, int which, size_t bytes, void **out);
// hipStreamSynchronize for a context with a communicator: polls the stream, RCCL's asynchronous error state and
// a deadline (PMF_COMM_TIMEOUT_S, default 1800; 0 = wait for ever), so that a peer that died or never arrived
// ends in PMF_ECOMM on the surviving ranks instead of a hang.
int pmf_comm_wait_stream(pmf_ctx *ctx, hipStream_t stream, const char *what);
// what a half-sweep's finalize writes (the arrays of `side` the SCATTER_GATHER exchange all-gathers) and whether
// PMF_EXCHANGE_AUTO should pick that exchange for it (true where finalize is expensive: the Gaussian row solves)
struct PmfExchange {
    bool prefer_scatter = false;
    int n_arrays = 0;
    int arrays[6] = {0, 0, 0, 0, 0, 0};
};
int pmf_comm_half_sweep(pmf_ctx *ctx, int side, size_t width, void *stats, bool chunked,
                        const std::function<int()> &accumulate, const std::function<int()> &finalize,
                        const PmfExchange &ex);

// profiling brackets (the *_on forms time work on another stream than the context's)
void pmf_prof_begin(pmf_ctx *ctx, int kernel);
void pmf_prof_end(pmf_ctx *ctx);
void pmf_prof_begin_on(pmf_ctx *ctx, int kernel, hipStream_t stream);
void pmf_prof_end_on(pmf_ctx *ctx, hipStream_t stream);

struct PmfProfScope {
    pmf_ctx *ctx;
    PmfProfScope(pmf_ctx *c, int kernel) : ctx(c) { pmf_prof_begin(c, kernel); }
    ~PmfProfScope() { pmf_prof_end(ctx); }
};

static inline int pmf_lanes_per_row(int kpad) {
    int kv = kpad / PMF_VEC;
    int l = 1;
    while (l < kv) l <<= 1;
    return l;
}
