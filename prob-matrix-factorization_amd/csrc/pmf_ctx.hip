// Context lifecycle, rating upload (stable CSR/CSC build + work lists), model
// state exchange and profiling for libpmf_hip.so.
#include <limits.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <new>

#include "pmf_internal.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void pmf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *pmf_last_error(void) { return g_err; }
extern "C" int pmf_abi_version(void) { return PMF_ABI_VERSION; }

extern "C" int pmf_device_count(int *count) {
    PMF_REQUIRE(count, PMF_EINVAL, "pmf_device_count: null argument");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        pmf_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return PMF_EHIP;
    }
    *count = n;
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------
int pmf_dev_alloc(pmf_ctx *ctx, void **p, size_t bytes) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        pmf_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? PMF_ENOMEM : PMF_EHIP;
    }
    ctx->device_bytes += (int64_t)bytes;
    return PMF_OK;
}

void pmf_dev_free(pmf_ctx *ctx, void *p, size_t bytes) {
    if (!p) return;
    (void)hipFree(p);
    ctx->device_bytes -= (int64_t)(bytes ? bytes : 16);
}

static int grow(pmf_ctx *ctx, void **p, size_t *have, size_t want) {
    if (*have >= want) return PMF_OK;
    if (*p) {
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        pmf_dev_free(ctx, *p, *have);
        *p = nullptr;
        *have = 0;
    }
    int rc = pmf_dev_alloc(ctx, p, want);
    if (rc) return rc;
    *have = want;
    return PMF_OK;
}

int pmf_ensure_partial(pmf_ctx *ctx, size_t bytes) {
    return grow(ctx, &ctx->d_partial, &ctx->partial_bytes, bytes);
}
int pmf_ensure_scratch(pmf_ctx *ctx, size_t bytes) {
    return grow(ctx, &ctx->d_scratch, &ctx->scratch_bytes, bytes);
}
int pmf_ensure_pinned(pmf_ctx *ctx, size_t bytes) {
    if (ctx->pinned_bytes >= bytes) return PMF_OK;
    if (ctx->h_pinned) {
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        (void)hipHostFree(ctx->h_pinned);
        ctx->h_pinned = nullptr;
        ctx->pinned_bytes = 0;
    }
    PMF_HIP_CHECK(hipHostMalloc(&ctx->h_pinned, bytes, hipHostMallocDefault));
    ctx->pinned_bytes = bytes;
    return PMF_OK;
}

size_t pmf_array_elems(const pmf_ctx *ctx, int side, int array) {
    size_t rows = (size_t)ctx->rows[side];
    switch (array) {
        case PMF_ARR_FACTOR:
        case PMF_ARR_SHAPE:
        case PMF_ARR_RATE:
            return rows * (size_t)ctx->kpad;
        case PMF_ARR_COV:
            return rows * (size_t)ctx->cov_stride;
        default:
            return rows;
    }
}

static const char *array_name(int array) {
    static const char *n[] = {"FACTOR", "SHAPE", "RATE", "PRIOR_RATE", "HYPER_RATE", "COV", "BIAS",
                              "SCALE", "SCALE_SHAPE", "SCALE_RATE"};
    return (array >= 0 && array < PMF_ARR_COUNT) ? n[array] : "?";
}

int pmf_require_array(pmf_ctx *ctx, int side, int array, const char *what) {
    PMF_REQUIRE(ctx->arr[side][array], PMF_EINVAL, "%s: array %s of side %d has not been set", what,
                array_name(array), side);
    return PMF_OK;
}

int pmf_alloc_array(pmf_ctx *ctx, int side, int array) {
    if (ctx->arr[side][array]) return PMF_OK;
    size_t bytes = pmf_array_elems(ctx, side, array) * ctx->elem;
    int rc = pmf_dev_alloc(ctx, &ctx->arr[side][array], bytes);
    if (rc) return rc;
    PMF_HIP_CHECK(hipMemsetAsync(ctx->arr[side][array], 0, bytes ? bytes : 16, ctx->stream));
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
#define CHECK_CTX(ctx, fn) PMF_REQUIRE((ctx) != nullptr, PMF_EINVAL, fn ": null context")
#define CHECK_SIDE(side, fn) \
    PMF_REQUIRE((side) == PMF_SIDE_USER || (side) == PMF_SIDE_ITEM, PMF_EINVAL, fn ": bad side %d", (side))

extern "C" int pmf_ctx_create(int device, int64_t n_users, int64_t n_items, int n_factors, int dtype,
                              pmf_ctx **out) {
    PMF_REQUIRE(out, PMF_EINVAL, "pmf_ctx_create: null out pointer");
    *out = nullptr;
    PMF_REQUIRE(n_users > 0 && n_items > 0, PMF_EINVAL, "pmf_ctx_create: dimensions must be positive");
    PMF_REQUIRE(n_users < INT32_MAX && n_items < INT32_MAX, PMF_ERANGE,
                "pmf_ctx_create: dimensions must fit int32");
    PMF_REQUIRE(n_factors >= 1 && n_factors <= 256, PMF_ERANGE,
                "pmf_ctx_create: n_factors=%d outside the supported range [1, 256]", n_factors);
    PMF_REQUIRE(dtype == PMF_F32 || dtype == PMF_F64, PMF_EINVAL, "pmf_ctx_create: bad dtype %d", dtype);
    int ndev = 0;
    PMF_HIP_CHECK(hipGetDeviceCount(&ndev));
    PMF_REQUIRE(device >= 0 && device < ndev, PMF_EINVAL, "pmf_ctx_create: device %d of %d", device, ndev);
    PMF_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    PMF_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    PMF_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, PMF_EINVAL,
                "pmf_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
    pmf_ctx *ctx = new (std::nothrow) pmf_ctx();
    PMF_REQUIRE(ctx, PMF_ENOMEM, "pmf_ctx_create: out of host memory");
    ctx->device = device;
    ctx->dtype = dtype;
    ctx->elem = dtype == PMF_F64 ? 8 : 4;
    ctx->rows[0] = n_users;
    ctx->rows[1] = n_items;
    ctx->K = n_factors;
    ctx->kpad = (n_factors + PMF_VEC - 1) / PMF_VEC * PMF_VEC;
    {
        // Factor rows are gathered at random: a row whose tail (row bytes mod 128) exceeds 64 bytes makes
        // at least every second gather touch one 128-byte line more than a line-aligned row would
        // (K = 20 fp32: 80-byte rows, 1.5 lines on average against 1).  Such rows are padded to the line.
        const int line = 128 / (int)ctx->elem;
        if ((ctx->kpad % line) * (int)ctx->elem > 64) ctx->kpad = (ctx->kpad + line - 1) / line * line;
    }
    ctx->kp = n_factors * (n_factors + 1) / 2;
    ctx->cov_stride = (ctx->kp + PMF_VEC - 1) / PMF_VEC * PMF_VEC;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        pmf_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return PMF_EHIP;
    }
    ctx->stream = ctx->own_stream;
    ctx->gauss_generic = getenv("PMF_GAUSS_GENERIC") != nullptr;
    ctx->gauss_unfused = getenv("PMF_GAUSS_UNFUSED") != nullptr;
    ctx->gauss_lds_solve = getenv("PMF_GAUSS_LDS_SOLVE") != nullptr;
    ctx->topk_two_phase = getenv("PMF_TOPK_TWO_PHASE") != nullptr;
    if (const char *nb = getenv("PMF_TOPK_STAGE_BUFFERS")) ctx->topk_stage_buffers = atoi(nb);
    if (const char *mb = getenv("PMF_TOPK_MAX_BLOCKS")) ctx->topk_max_blocks = atoi(mb);
    if (const char *ex = getenv("PMF_COMM_EXCHANGE")) {
        if (!strcmp(ex, "allreduce")) ctx->exchange = PMF_EXCHANGE_ALLREDUCE;
        else if (!strcmp(ex, "scatter_gather")) ctx->exchange = PMF_EXCHANGE_SCATTER_GATHER;
    }
    *out = ctx;
    return PMF_OK;
}

static void free_tasks(pmf_ctx *ctx, PmfTaskList &t) {
    pmf_dev_free(ctx, t.d_tasks, (size_t)t.n_tasks * sizeof(PmfTask));
    pmf_dev_free(ctx, t.d_split, (size_t)t.n_split * sizeof(PmfSplitRow));
    pmf_dev_free(ctx, t.d_split_rows, (size_t)t.n_split * sizeof(int32_t));
    t = PmfTaskList();
}

static void free_index(pmf_ctx *ctx) {
    for (int s = 0; s < 2; ++s) {
        PmfSideIndex &ix = ctx->index[s];
        pmf_dev_free(ctx, ix.d_ptr, (size_t)(ctx->rows[s] + 1) * sizeof(int64_t));
        pmf_dev_free(ctx, ix.d_other, (size_t)ctx->nnz * sizeof(int32_t));
        pmf_dev_free(ctx, ix.d_val, (size_t)ctx->nnz * ctx->elem);
        ix.d_ptr = nullptr;
        ix.d_other = nullptr;
        ix.d_val = nullptr;
        pmf_dev_free(ctx, ix.d_nonempty, (size_t)ix.n_nonempty * sizeof(int32_t));
        ix.d_nonempty = nullptr;
        ix.n_nonempty = 0;
        ix.h_ptr.clear();
        ix.h_nonempty.clear();
        ix.nonempty_off.clear();
        free_tasks(ctx, ix.gamma_tasks);
        free_tasks(ctx, ix.gauss_tasks);
        free_tasks(ctx, ix.bias_tasks);
        free_tasks(ctx, ix.sgd_tasks);
    }
    ctx->nnz = 0;
}

static void free_eval(pmf_ctx *ctx) {
    PmfEvalSet &ev = ctx->eval;
    pmf_dev_free(ctx, ev.d_u, (size_t)ev.n * 4);
    pmf_dev_free(ctx, ev.d_i, (size_t)ev.n * 4);
    pmf_dev_free(ctx, ev.d_y, (size_t)ev.n * 8);
    pmf_dev_free(ctx, ev.d_label, (size_t)ev.n * 4);
    ev = PmfEvalSet();
}

extern "C" int pmf_ctx_destroy(pmf_ctx *ctx) {
    if (!ctx) return PMF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    pmf_comm_release(ctx);
    for (auto &r : ctx->prof_pending) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (auto &e : ctx->prof_pool) (void)hipEventDestroy(e);
    free_index(ctx);
    free_eval(ctx);
    for (int s = 0; s < 2; ++s)
        for (int a = 0; a < PMF_ARR_COUNT; ++a)
            if (ctx->arr[s][a]) pmf_dev_free(ctx, ctx->arr[s][a], pmf_array_elems(ctx, s, a) * ctx->elem);
    pmf_dev_free(ctx, ctx->d_partial, ctx->partial_bytes);
    pmf_dev_free(ctx, ctx->d_scratch, ctx->scratch_bytes);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return PMF_OK;
}

extern "C" int pmf_ctx_set_stream(pmf_ctx *ctx, void *hip_stream) {
    CHECK_CTX(ctx, "pmf_ctx_set_stream");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PMF_OK;
}

extern "C" int pmf_ctx_sync(pmf_ctx *ctx) {
    CHECK_CTX(ctx, "pmf_ctx_sync");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    if (pmf_comm_active(ctx)) return pmf_comm_wait_stream(ctx, ctx->stream, "pmf_ctx_sync");   // (watches the peers too)
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PMF_OK;
}

extern "C" int pmf_ctx_device_bytes(pmf_ctx *ctx, int64_t *bytes) {
    CHECK_CTX(ctx, "pmf_ctx_device_bytes");
    PMF_REQUIRE(bytes, PMF_EINVAL, "pmf_ctx_device_bytes: null argument");
    *bytes = ctx->device_bytes;
    return PMF_OK;
}

extern "C" int pmf_ctx_kpad(pmf_ctx *ctx, int *kpad) {
    CHECK_CTX(ctx, "pmf_ctx_kpad");
    PMF_REQUIRE(kpad, PMF_EINVAL, "pmf_ctx_kpad: null argument");
    *kpad = ctx->kpad;
    return PMF_OK;
}

extern "C" int pmf_ctx_cov_stride(pmf_ctx *ctx, int *stride) {
    CHECK_CTX(ctx, "pmf_ctx_cov_stride");
    PMF_REQUIRE(stride, PMF_EINVAL, "pmf_ctx_cov_stride: null argument");
    *stride = ctx->cov_stride;
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// ratings: stable counting sort by row + work lists
// ---------------------------------------------------------------------------
// Cut every row into runs of at most `chunk` ratings.  Rows that fit one run
// become self-contained tasks; longer rows are split evenly and listed in
// `split` so a second kernel can combine their partial sums in slot order.
// Tasks are emitted longest-first so that the lane groups of one wavefront
// (which each take one task) carry similar amounts of work and the grid's tail
// consists of short tasks.
//
// With row chunks (pmf_ctx_set_row_chunks) the tasks are grouped by the chunk of
// their row -- longest-first inside every group -- and `task_off` / `split_off`
// hold the group boundaries (the split list is in row order, hence grouped too).
static void build_tasks(const std::vector<int64_t> &ptr, int64_t rows, int chunk, bool keep_empty,
                        const std::vector<int64_t> &row_bounds, std::vector<PmfTask> &tasks,
                        std::vector<PmfSplitRow> &split, int64_t &n_slots, std::vector<int64_t> &task_off,
                        std::vector<int64_t> &split_off) {
    std::vector<PmfTask> raw;
    raw.reserve((size_t)rows + 1024);
    n_slots = 0;
    for (int64_t r = 0; r < rows; ++r) {
        int64_t n = ptr[r + 1] - ptr[r];
        if (n == 0 && !keep_empty) continue;
        if (n <= chunk) {
            raw.push_back(PmfTask{ptr[r], (int32_t)r, (int32_t)n, -1, 0});
            continue;
        }
        int64_t q = (n + chunk - 1) / chunk;
        int64_t base = n / q, extra = n % q, at = ptr[r];
        split.push_back(PmfSplitRow{(int32_t)r, (int32_t)n_slots, (int32_t)q, 0});
        for (int64_t c = 0; c < q; ++c) {
            int64_t len = base + (c < extra ? 1 : 0);
            raw.push_back(PmfTask{at, (int32_t)r, (int32_t)len, (int32_t)(n_slots + c), 0});
            at += len;
        }
        n_slots += q;
    }
    // counting sort by (row chunk, length descending), stable; `raw` is in row order
    const size_t n_groups = row_bounds.size() - 1, span = (size_t)chunk + 1;
    std::vector<int64_t> bucket(n_groups * span + 1, 0);
    std::vector<int32_t> group(raw.size());
    {
        size_t g = 0;
        for (size_t k = 0; k < raw.size(); ++k) {
            while (raw[k].row >= row_bounds[g + 1]) ++g;
            group[k] = (int32_t)g;
            bucket[g * span + (size_t)(chunk - raw[k].len) + 1]++;
        }
    }
    for (size_t b = 1; b < bucket.size(); ++b) bucket[b] += bucket[b - 1];
    task_off.resize(n_groups + 1);
    for (size_t g = 0; g <= n_groups; ++g) task_off[g] = bucket[g * span];
    tasks.resize(raw.size());
    for (size_t k = 0; k < raw.size(); ++k)
        tasks[(size_t)bucket[(size_t)group[k] * span + (size_t)(chunk - raw[k].len)]++] = raw[k];
    split_off.assign(n_groups + 1, (int64_t)split.size());
    {
        size_t k = 0;
        for (size_t g = 0; g < n_groups; ++g) {
            while (k < split.size() && split[k].row < row_bounds[g]) ++k;
            split_off[g] = (int64_t)k;
        }
    }
}

static std::vector<int64_t> chunk_bounds(const pmf_ctx *ctx, int side) {
    std::vector<int64_t> b((size_t)ctx->n_chunks[side] + 1);
    for (int c = 0; c <= ctx->n_chunks[side]; ++c) b[(size_t)c] = pmf_chunk_row0(ctx, side, c);
    return b;
}

PmfTaskView pmf_task_view(const pmf_ctx *ctx, int side, const PmfTaskList &tl, bool select) {
    const PmfSideIndex &ix = ctx->index[side];
    const int c = select ? ctx->cur_chunk[side] : -1;
    PmfTaskView v;
    v.n_slots = tl.n_slots;
    // a finalize window (pmf_comm_half_sweep, SCATTER_GATHER exchange) narrows the ROW RANGE of the view; the
    // finalize-from-statistics launches use nothing else of it
    auto clip = [&](PmfTaskView &w) {
        if (select && ctx->fin_row0 >= 0) {
            w.row0 = std::max(w.row0, ctx->fin_row0);
            w.row1 = std::max(w.row0, std::min(w.row1, ctx->fin_row1));
        }
    };
    if (c < 0 || tl.task_off.empty()) {
        v.d_tasks = tl.d_tasks;
        v.d_split = tl.d_split;
        v.d_split_rows = tl.d_split_rows;
        v.n_tasks = tl.n_tasks;
        v.n_split = tl.n_split;
        v.row0 = 0;
        v.row1 = ctx->rows[side];
        v.d_nonempty = ix.d_nonempty;
        v.n_nonempty = ix.n_nonempty;
        clip(v);
        return v;
    }
    const size_t g = (size_t)c;
    v.d_tasks = tl.d_tasks + tl.task_off[g];
    v.n_tasks = tl.task_off[g + 1] - tl.task_off[g];
    v.d_split = tl.d_split + tl.split_off[g];
    v.d_split_rows = tl.d_split_rows + tl.split_off[g];
    v.n_split = tl.split_off[g + 1] - tl.split_off[g];
    v.row0 = pmf_chunk_row0(ctx, side, c);
    v.row1 = pmf_chunk_row0(ctx, side, c + 1);
    v.d_nonempty = ix.d_nonempty + ix.nonempty_off[g];
    v.n_nonempty = ix.nonempty_off[g + 1] - ix.nonempty_off[g];
    clip(v);
    return v;
}

static int upload_tasks(pmf_ctx *ctx, int side, const std::vector<int64_t> &ptr, int64_t rows, int chunk,
                        bool keep_empty, PmfTaskList &out) {
    std::vector<PmfTask> tasks;
    std::vector<PmfSplitRow> split;
    int64_t n_slots = 0;
    build_tasks(ptr, rows, chunk, keep_empty, chunk_bounds(ctx, side), tasks, split, n_slots, out.task_off,
                out.split_off);
    out.n_tasks = (int64_t)tasks.size();
    out.n_split = (int64_t)split.size();
    out.n_slots = n_slots;
    out.max_len = tasks.empty() ? 0 : tasks.front().len;
    int rc = pmf_dev_alloc(ctx, (void **)&out.d_tasks, tasks.size() * sizeof(PmfTask));
    if (rc) return rc;
    rc = pmf_dev_alloc(ctx, (void **)&out.d_split, split.size() * sizeof(PmfSplitRow));
    if (rc) return rc;
    if (!tasks.empty())
        PMF_HIP_CHECK(hipMemcpy(out.d_tasks, tasks.data(), tasks.size() * sizeof(PmfTask), hipMemcpyHostToDevice));
    rc = pmf_dev_alloc(ctx, (void **)&out.d_split_rows, split.size() * sizeof(int32_t));
    if (rc) return rc;
    if (!split.empty()) {
        PMF_HIP_CHECK(hipMemcpy(out.d_split, split.data(), split.size() * sizeof(PmfSplitRow), hipMemcpyHostToDevice));
        std::vector<int32_t> ids(split.size());
        for (size_t k = 0; k < split.size(); ++k) ids[k] = split[k].row;
        PMF_HIP_CHECK(hipMemcpy(out.d_split_rows, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return PMF_OK;
}

// (re)build the three task lists of one side for the current row chunking
static int build_work_lists(pmf_ctx *ctx, int side) {
    PmfSideIndex &ix = ctx->index[side];
    const int64_t rows = ctx->rows[side];
    free_tasks(ctx, ix.gamma_tasks);
    free_tasks(ctx, ix.gauss_tasks);
    free_tasks(ctx, ix.bias_tasks);
    free_tasks(ctx, ix.sgd_tasks);
    const std::vector<int64_t> bounds = chunk_bounds(ctx, side);
    ix.nonempty_off.assign(bounds.size(), (int64_t)ix.h_nonempty.size());
    for (size_t g = 0; g + 1 < bounds.size(); ++g)
        ix.nonempty_off[g] = std::lower_bound(ix.h_nonempty.begin(), ix.h_nonempty.end(), (int32_t)bounds[g]) -
                             ix.h_nonempty.begin();
    // A task is walked sequentially by one lane group / wavefront, so the longest task bounds the
    // launch from below (C1: 256-rating tasks = 16 dependent gather rounds = 29 us for a 200k-rating
    // sweep).  Small problems therefore get shorter tasks -- enough of them to occupy the chip -- and
    // large ones keep the maximum, which minimises partial-sum traffic.  (Chunking only changes the
    // summation order of rows longer than a chunk.)  The gradient mode has its own list with the fixed
    // 256: its result is defined in terms of that piece length.
    auto task_chunk = [&](int max_chunk) {
        int64_t c = 32;
        while (c < max_chunk && c * 65536 < ctx->nnz) c <<= 1;
        return (int)c;
    };
    int rc;
    if ((rc = upload_tasks(ctx, side, ix.h_ptr, rows, task_chunk(PMF_GAMMA_CHUNK), true, ix.gamma_tasks))) return rc;
    if ((rc = upload_tasks(ctx, side, ix.h_ptr, rows, task_chunk(PMF_GAUSS_CHUNK), false, ix.gauss_tasks))) return rc;
    if ((rc = upload_tasks(ctx, side, ix.h_ptr, rows, task_chunk(PMF_GAMMA_CHUNK), false, ix.bias_tasks))) return rc;
    if ((rc = upload_tasks(ctx, side, ix.h_ptr, rows, PMF_SGD_CHUNK, false, ix.sgd_tasks))) return rc;
    return PMF_OK;
}

static int set_ratings_impl(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids, const int32_t *item_ids,
                            const double *ratings);

extern "C" int pmf_ctx_set_ratings(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids,
                                   const int32_t *item_ids, const double *ratings) {
    try {  // host containers may throw: nothing propagates across the C boundary
        return set_ratings_impl(ctx, nnz, user_ids, item_ids, ratings);
    } catch (const std::bad_alloc &) {
        pmf_set_error("pmf_ctx_set_ratings: out of host memory");
        return PMF_ENOMEM;
    } catch (const std::exception &e) {
        pmf_set_error("pmf_ctx_set_ratings: %s", e.what());
        return PMF_EINVAL;
    }
}

// Host build of one side (stable counting sort).  Used for nnz >= 2^31 - 1 and with
// PMF_INDEX_HOST=1 (the tests compare it with the device build of pmf_index.hip).
static int fill_side_host(pmf_ctx *ctx, int side, int64_t nnz, const int32_t *key, const int32_t *oth,
                          const double *ratings, std::vector<int32_t> &other, std::vector<char> &val) {
    const int64_t rows = ctx->rows[side];
    PmfSideIndex &ix = ctx->index[side];
    ix.h_ptr.assign((size_t)rows + 1, 0);
    for (int64_t n = 0; n < nnz; ++n) ix.h_ptr[(size_t)key[n] + 1]++;
    for (int64_t r = 0; r < rows; ++r) ix.h_ptr[(size_t)r + 1] += ix.h_ptr[(size_t)r];
    std::vector<int64_t> cursor(ix.h_ptr.begin(), ix.h_ptr.end() - 1);
    if (ctx->dtype == PMF_F64) {
        double *v = (double *)val.data();
        for (int64_t n = 0; n < nnz; ++n) {
            int64_t d = cursor[(size_t)key[n]]++;
            other[(size_t)d] = oth[n];
            v[d] = ratings[n];
        }
    } else {
        float *v = (float *)val.data();
        for (int64_t n = 0; n < nnz; ++n) {
            int64_t d = cursor[(size_t)key[n]]++;
            other[(size_t)d] = oth[n];
            v[d] = (float)ratings[n];
        }
    }
    PMF_HIP_CHECK(hipMemcpy(ix.d_ptr, ix.h_ptr.data(), (size_t)(rows + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
    if (nnz) {
        PMF_HIP_CHECK(hipMemcpy(ix.d_other, other.data(), (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice));
        PMF_HIP_CHECK(hipMemcpy(ix.d_val, val.data(), (size_t)nnz * ctx->elem, hipMemcpyHostToDevice));
    }
    return PMF_OK;
}

static int bad_id_error(const pmf_ctx *ctx, const int32_t *user_ids, const int32_t *item_ids, int64_t n) {
    const int64_t U = ctx->rows[0], I = ctx->rows[1];
    PMF_REQUIRE(user_ids[n] >= 0 && user_ids[n] < U, PMF_ERANGE,
                "pmf_ctx_set_ratings: user id %d at position %lld outside [0, %lld)", user_ids[n], (long long)n,
                (long long)U);
    PMF_REQUIRE(item_ids[n] >= 0 && item_ids[n] < I, PMF_ERANGE,
                "pmf_ctx_set_ratings: item id %d at position %lld outside [0, %lld)", item_ids[n], (long long)n,
                (long long)I);
    return PMF_OK;
}

static int set_ratings_impl(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids, const int32_t *item_ids,
                            const double *ratings) {
    CHECK_CTX(ctx, "pmf_ctx_set_ratings");
    PMF_REQUIRE(nnz >= 0, PMF_EINVAL, "pmf_ctx_set_ratings: negative nnz");
    PMF_REQUIRE(nnz == 0 || (user_ids && item_ids && ratings), PMF_EINVAL,
                "pmf_ctx_set_ratings: null input array");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // positions are sorted as 32-bit values on the device
    const bool on_host = nnz >= (int64_t)INT32_MAX || getenv("PMF_INDEX_HOST") != nullptr;
    int rc;
    PmfIndexBuild *build = nullptr;
    // validation comes first: a bad id leaves the context's previous ratings in place
    if (on_host) {
        for (int64_t n = 0; n < nnz; ++n)
            if ((rc = bad_id_error(ctx, user_ids, item_ids, n))) return rc;
    } else {
        int64_t bad = -1;
        if ((rc = pmf_index_device_begin(ctx, nnz, user_ids, item_ids, ratings, &build, &bad))) return rc;
        if (bad >= 0) return bad_id_error(ctx, user_ids, item_ids, bad);
    }
    free_index(ctx);
    ctx->nnz = nnz;
    for (int side = 0; side < 2; ++side) {
        PmfSideIndex &ix = ctx->index[side];
        rc = pmf_dev_alloc(ctx, (void **)&ix.d_ptr, (size_t)(ctx->rows[side] + 1) * sizeof(int64_t));
        if (!rc) rc = pmf_dev_alloc(ctx, (void **)&ix.d_other, (size_t)nnz * sizeof(int32_t));
        if (!rc) rc = pmf_dev_alloc(ctx, &ix.d_val, (size_t)nnz * ctx->elem);
        if (rc) {
            pmf_index_device_abort(build);
            return rc;
        }
    }
    if (on_host) {
        std::vector<int32_t> other((size_t)nnz);
        std::vector<char> val((size_t)nnz * ctx->elem);
        if ((rc = fill_side_host(ctx, PMF_SIDE_USER, nnz, user_ids, item_ids, ratings, other, val))) return rc;
        if ((rc = fill_side_host(ctx, PMF_SIDE_ITEM, nnz, item_ids, user_ids, ratings, other, val))) return rc;
    } else {
        if ((rc = pmf_index_device_finish(ctx, build, nnz))) return rc;  // consumes `build`
    }
    for (int side = 0; side < 2; ++side) {
        const int64_t rows = ctx->rows[side];
        PmfSideIndex &ix = ctx->index[side];
        std::vector<int32_t> nonempty;
        for (int64_t r = 0; r < rows; ++r)
            if (ix.h_ptr[(size_t)r + 1] > ix.h_ptr[(size_t)r]) nonempty.push_back((int32_t)r);
        ix.n_nonempty = (int64_t)nonempty.size();
        if ((rc = pmf_dev_alloc(ctx, (void **)&ix.d_nonempty, nonempty.size() * sizeof(int32_t)))) return rc;
        if (!nonempty.empty())
            PMF_HIP_CHECK(hipMemcpy(ix.d_nonempty, nonempty.data(), nonempty.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        ix.h_nonempty.swap(nonempty);
        if ((rc = build_work_lists(ctx, side))) return rc;
    }
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// row chunks: a half-sweep split into row ranges so that a multi-GPU caller can
// all-reduce the statistics of one range while the next one is being accumulated
// ---------------------------------------------------------------------------
static int set_row_chunks_impl(pmf_ctx *ctx, int side, int n_chunks) {
    CHECK_CTX(ctx, "pmf_ctx_set_row_chunks");
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, "pmf_ctx_set_row_chunks: bad side %d", side);
    PMF_REQUIRE(n_chunks >= 1 && n_chunks <= 1024, PMF_EINVAL,
                "pmf_ctx_set_row_chunks: n_chunks = %d outside [1, 1024]", n_chunks);
    if ((int64_t)n_chunks > ctx->rows[side]) n_chunks = (int)(ctx->rows[side] > 0 ? ctx->rows[side] : 1);
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->n_chunks[side] = n_chunks;
    ctx->cur_chunk[side] = -1;
    if (ctx->index[side].d_ptr) return build_work_lists(ctx, side);
    return PMF_OK;
}

extern "C" int pmf_ctx_set_row_chunks(pmf_ctx *ctx, int side, int n_chunks) {
    try {
        return set_row_chunks_impl(ctx, side, n_chunks);
    } catch (const std::bad_alloc &) {
        pmf_set_error("pmf_ctx_set_row_chunks: out of host memory");
        return PMF_ENOMEM;
    }
}

extern "C" int pmf_ctx_chunk_rows(pmf_ctx *ctx, int side, int chunk, int64_t *row_begin, int64_t *row_end) {
    CHECK_CTX(ctx, "pmf_ctx_chunk_rows");
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, "pmf_ctx_chunk_rows: bad side %d", side);
    PMF_REQUIRE(row_begin && row_end, PMF_EINVAL, "pmf_ctx_chunk_rows: null argument");
    PMF_REQUIRE(chunk >= 0 && chunk < ctx->n_chunks[side], PMF_ERANGE,
                "pmf_ctx_chunk_rows: chunk %d outside [0, %d)", chunk, ctx->n_chunks[side]);
    *row_begin = pmf_chunk_row0(ctx, side, chunk);
    *row_end = pmf_chunk_row0(ctx, side, chunk + 1);
    return PMF_OK;
}

extern "C" int pmf_ctx_select_chunk(pmf_ctx *ctx, int side, int chunk) {
    CHECK_CTX(ctx, "pmf_ctx_select_chunk");
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, "pmf_ctx_select_chunk: bad side %d", side);
    PMF_REQUIRE(chunk >= -1 && chunk < ctx->n_chunks[side], PMF_ERANGE,
                "pmf_ctx_select_chunk: chunk %d outside [-1, %d)", chunk, ctx->n_chunks[side]);
    ctx->cur_chunk[side] = chunk;
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// model state exchange (host float64 <-> device dtype, padded / packed layouts)
// ---------------------------------------------------------------------------
template <typename T>
static void pack_rows(const double *src, T *dst, int64_t rows, int width, int stride) {
    for (int64_t r = 0; r < rows; ++r) {
        const double *s = src + r * width;
        T *d = dst + r * stride;
        for (int k = 0; k < width; ++k) d[k] = (T)s[k];
        for (int k = width; k < stride; ++k) d[k] = (T)0;
    }
}

template <typename T>
static void unpack_rows(const T *src, double *dst, int64_t rows, int width, int stride) {
    for (int64_t r = 0; r < rows; ++r) {
        const T *s = src + r * stride;
        double *d = dst + r * width;
        for (int k = 0; k < width; ++k) d[k] = (double)s[k];
    }
}

// full K x K (host) -> packed lower triangle, row-major: (r, c), c <= r at r(r+1)/2 + c
template <typename T>
static void pack_cov(const double *src, T *dst, int64_t rows, int K, int stride) {
    for (int64_t n = 0; n < rows; ++n) {
        const double *s = src + n * (int64_t)K * K;
        T *d = dst + n * stride;
        int p = 0;
        for (int r = 0; r < K; ++r)
            for (int c = 0; c <= r; ++c) d[p++] = (T)s[r * K + c];
        for (; p < stride; ++p) d[p] = (T)0;
    }
}

template <typename T>
static void unpack_cov(const T *src, double *dst, int64_t rows, int K, int stride) {
    for (int64_t n = 0; n < rows; ++n) {
        const T *s = src + n * stride;
        double *d = dst + n * (int64_t)K * K;
        int p = 0;
        for (int r = 0; r < K; ++r)
            for (int c = 0; c <= r; ++c) {
                double v = (double)s[p++];
                d[r * K + c] = v;
                d[c * K + r] = v;
            }
    }
}

void pmf_array_shape(const pmf_ctx *ctx, int array, int *host_width, int *dev_stride) {
    switch (array) {
        case PMF_ARR_FACTOR:
        case PMF_ARR_SHAPE:
        case PMF_ARR_RATE:
            *host_width = ctx->K;
            *dev_stride = ctx->kpad;
            break;
        case PMF_ARR_COV:
            *host_width = ctx->K * ctx->K;
            *dev_stride = ctx->cov_stride;
            break;
        default:
            *host_width = 1;
            *dev_stride = 1;
    }
}

void pmf_unpack_rows(const pmf_ctx *ctx, int array, const void *src, double *dst, int64_t rows) {
    int width, stride;
    pmf_array_shape(ctx, array, &width, &stride);
    if (array == PMF_ARR_COV) {
        if (ctx->dtype == PMF_F64) unpack_cov((const double *)src, dst, rows, ctx->K, stride);
        else unpack_cov((const float *)src, dst, rows, ctx->K, stride);
    } else {
        if (ctx->dtype == PMF_F64) unpack_rows((const double *)src, dst, rows, width, stride);
        else unpack_rows((const float *)src, dst, rows, width, stride);
    }
}

static const int64_t kStageBytes = 64ll << 20;

extern "C" int pmf_set_array(pmf_ctx *ctx, int side, int array, const double *host) {
    CHECK_CTX(ctx, "pmf_set_array");
    CHECK_SIDE(side, "pmf_set_array");
    PMF_REQUIRE(array >= 0 && array < PMF_ARR_COUNT, PMF_EINVAL, "pmf_set_array: bad array id %d", array);
    PMF_REQUIRE(host, PMF_EINVAL, "pmf_set_array: null host pointer");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = pmf_alloc_array(ctx, side, array);
    if (rc) return rc;
    int width, stride;
    pmf_array_shape(ctx, array, &width, &stride);
    const int64_t rows = ctx->rows[side];
    const int64_t row_bytes = (int64_t)stride * (int64_t)ctx->elem;
    const int64_t step = std::max<int64_t>(1, kStageBytes / row_bytes);
    if ((rc = pmf_ensure_pinned(ctx, (size_t)(std::min(step, rows) * row_bytes)))) return rc;
    for (int64_t r0 = 0; r0 < rows; r0 += step) {
        int64_t nr = std::min(step, rows - r0);
        const double *src = host + r0 * width;
        if (array == PMF_ARR_COV) {
            if (ctx->dtype == PMF_F64) pack_cov(src, (double *)ctx->h_pinned, nr, ctx->K, stride);
            else pack_cov(src, (float *)ctx->h_pinned, nr, ctx->K, stride);
        } else {
            if (ctx->dtype == PMF_F64) pack_rows(src, (double *)ctx->h_pinned, nr, width, stride);
            else pack_rows(src, (float *)ctx->h_pinned, nr, width, stride);
        }
        char *dst = (char *)ctx->arr[side][array] + r0 * row_bytes;
        PMF_HIP_CHECK(hipMemcpyAsync(dst, ctx->h_pinned, (size_t)(nr * row_bytes), hipMemcpyHostToDevice, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PMF_OK;
}

extern "C" int pmf_get_array(pmf_ctx *ctx, int side, int array, double *host) {
    CHECK_CTX(ctx, "pmf_get_array");
    CHECK_SIDE(side, "pmf_get_array");
    PMF_REQUIRE(array >= 0 && array < PMF_ARR_COUNT, PMF_EINVAL, "pmf_get_array: bad array id %d", array);
    PMF_REQUIRE(host, PMF_EINVAL, "pmf_get_array: null host pointer");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = pmf_require_array(ctx, side, array, "pmf_get_array");
    if (rc) return rc;
    int width, stride;
    pmf_array_shape(ctx, array, &width, &stride);
    const int64_t rows = ctx->rows[side];
    const int64_t row_bytes = (int64_t)stride * (int64_t)ctx->elem;
    const int64_t step = std::max<int64_t>(1, kStageBytes / row_bytes);
    if ((rc = pmf_ensure_pinned(ctx, (size_t)(std::min(step, rows) * row_bytes)))) return rc;
    for (int64_t r0 = 0; r0 < rows; r0 += step) {
        int64_t nr = std::min(step, rows - r0);
        const char *src = (const char *)ctx->arr[side][array] + r0 * row_bytes;
        PMF_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, src, (size_t)(nr * row_bytes), hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        pmf_unpack_rows(ctx, array, ctx->h_pinned, host + r0 * width, nr);
    }
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// row-subset exchange: the reference's callers index rows of the state (V_theta[i], m_beta[j_idx]:
// gaussian_mf_cavi_bias.py:146-162); at 1M+ rows a K x K covariance stack is tens of GB as host float64,
// so a caller that wants a few rows gets a few rows.  Rows are gathered / scattered on the device in
// units of 16 bytes (per-row scalars: one element), 64-bit offsets throughout.
// ---------------------------------------------------------------------------
template <typename U, bool SCATTER>
__global__ void rows_copy_kernel(U *table, U *staged, const int64_t *rows, int64_t n, int64_t units_per_row) {
    const int64_t total = n * units_per_row;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / units_per_row, w = t - r * units_per_row;
        U *at = table + rows[r] * units_per_row + w;
        if (SCATTER) *at = staged[t];
        else staged[t] = *at;
    }
}

template <bool SCATTER>
static void launch_rows_copy(pmf_ctx *ctx, void *table, void *staged, const int64_t *d_rows, int64_t n, int64_t row_bytes) {
    const bool wide = row_bytes % 16 == 0;
    const int64_t unit = wide ? 16 : (int64_t)ctx->elem, upr = row_bytes / unit;
    const unsigned grid = (unsigned)std::min<int64_t>((n * upr + 255) / 256, 65536);
    if (wide)
        hipLaunchKernelGGL((rows_copy_kernel<uint4, SCATTER>), dim3(grid), dim3(256), 0, ctx->stream, (uint4 *)table,
                           (uint4 *)staged, d_rows, n, upr);
    else if (unit == 8)
        hipLaunchKernelGGL((rows_copy_kernel<uint2, SCATTER>), dim3(grid), dim3(256), 0, ctx->stream, (uint2 *)table,
                           (uint2 *)staged, d_rows, n, upr);
    else
        hipLaunchKernelGGL((rows_copy_kernel<uint32_t, SCATTER>), dim3(grid), dim3(256), 0, ctx->stream, (uint32_t *)table,
                           (uint32_t *)staged, d_rows, n, upr);
}

// shared body of pmf_get_array_rows (host_out) / pmf_set_array_rows (host_in)
static int array_rows_impl(pmf_ctx *ctx, int side, int array, int64_t n, const int64_t *rows, const double *host_in,
                           double *host_out, const char *fn) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "%s: null context", fn);
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, "%s: bad side %d", fn, side);
    PMF_REQUIRE(array >= 0 && array < PMF_ARR_COUNT, PMF_EINVAL, "%s: bad array id %d", fn, array);
    PMF_REQUIRE(n >= 0, PMF_EINVAL, "%s: negative row count", fn);
    if (n == 0) return PMF_OK;
    PMF_REQUIRE(rows && (host_in || host_out), PMF_EINVAL, "%s: null argument", fn);
    for (int64_t k = 0; k < n; ++k)
        PMF_REQUIRE(rows[k] >= 0 && rows[k] < ctx->rows[side], PMF_ERANGE, "%s: row %lld at position %lld outside [0, %lld)",
                    fn, (long long)rows[k], (long long)k, (long long)ctx->rows[side]);
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = host_in ? pmf_alloc_array(ctx, side, array) : pmf_require_array(ctx, side, array, fn);
    if (rc) return rc;
    int width, stride;
    pmf_array_shape(ctx, array, &width, &stride);
    const int64_t row_bytes = (int64_t)stride * (int64_t)ctx->elem;
    const int64_t step = std::min(n, std::max<int64_t>(1, kStageBytes / row_bytes));
    // device staging: [step rows of the array's layout][step row ids]; the host side of it is the pinned buffer
    const size_t data_bytes = (size_t)((step * row_bytes + 15) / 16 * 16);
    if ((rc = pmf_ensure_scratch(ctx, data_bytes + (size_t)step * sizeof(int64_t)))) return rc;
    if ((rc = pmf_ensure_pinned(ctx, std::max(data_bytes, (size_t)step * sizeof(int64_t))))) return rc;
    char *d_data = (char *)ctx->d_scratch;
    int64_t *d_rows = (int64_t *)(d_data + data_bytes);
    for (int64_t r0 = 0; r0 < n; r0 += step) {
        const int64_t nr = std::min(step, n - r0);
        memcpy(ctx->h_pinned, rows + r0, (size_t)nr * sizeof(int64_t));
        PMF_HIP_CHECK(hipMemcpyAsync(d_rows, ctx->h_pinned, (size_t)nr * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // the pinned buffer is reused below
        if (host_in) {
            const double *src = host_in + r0 * width;
            if (array == PMF_ARR_COV) {
                if (ctx->dtype == PMF_F64) pack_cov(src, (double *)ctx->h_pinned, nr, ctx->K, stride);
                else pack_cov(src, (float *)ctx->h_pinned, nr, ctx->K, stride);
            } else {
                if (ctx->dtype == PMF_F64) pack_rows(src, (double *)ctx->h_pinned, nr, width, stride);
                else pack_rows(src, (float *)ctx->h_pinned, nr, width, stride);
            }
            PMF_HIP_CHECK(hipMemcpyAsync(d_data, ctx->h_pinned, (size_t)(nr * row_bytes), hipMemcpyHostToDevice, ctx->stream));
            launch_rows_copy<true>(ctx, ctx->arr[side][array], d_data, d_rows, nr, row_bytes);
            PMF_HIP_CHECK(hipGetLastError());
            PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        } else {
            launch_rows_copy<false>(ctx, ctx->arr[side][array], d_data, d_rows, nr, row_bytes);
            PMF_HIP_CHECK(hipGetLastError());
            PMF_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, d_data, (size_t)(nr * row_bytes), hipMemcpyDeviceToHost, ctx->stream));
            PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            pmf_unpack_rows(ctx, array, ctx->h_pinned, host_out + r0 * width, nr);
        }
    }
    return PMF_OK;
}

extern "C" int pmf_get_array_rows(pmf_ctx *ctx, int side, int array, int64_t n, const int64_t *rows, double *host) {
    return array_rows_impl(ctx, side, array, n, rows, nullptr, host, "pmf_get_array_rows");
}

extern "C" int pmf_set_array_rows(pmf_ctx *ctx, int side, int array, int64_t n, const int64_t *rows, const double *host) {
    return array_rows_impl(ctx, side, array, n, rows, host, nullptr, "pmf_set_array_rows");
}

template <typename T>
__global__ void cov_identity_kernel(T *cov, int64_t rows, int K, int stride, T scale) {
    int64_t total = rows * stride;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int p = (int)(t % stride);
        // p is a diagonal entry iff p = r(r+3)/2 for some r < K
        int r = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
        while ((r + 1) * (r + 2) / 2 <= p) ++r;
        while (r * (r + 1) / 2 > p) --r;
        bool diag = (p - r * (r + 1) / 2 == r) && r < K;
        cov[t] = diag ? scale : (T)0;
    }
}

extern "C" int pmf_set_cov_identity(pmf_ctx *ctx, int side, double scale) {
    CHECK_CTX(ctx, "pmf_set_cov_identity");
    CHECK_SIDE(side, "pmf_set_cov_identity");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = pmf_alloc_array(ctx, side, PMF_ARR_COV);
    if (rc) return rc;
    int64_t total = ctx->rows[side] * ctx->cov_stride;
    int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
    if (ctx->dtype == PMF_F64)
        hipLaunchKernelGGL(cov_identity_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream,
                           (double *)ctx->arr[side][PMF_ARR_COV], ctx->rows[side], ctx->K, ctx->cov_stride, scale);
    else
        hipLaunchKernelGGL(cov_identity_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream,
                           (float *)ctx->arr[side][PMF_ARR_COV], ctx->rows[side], ctx->K, ctx->cov_stride, (float)scale);
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
static hipEvent_t take_event(pmf_ctx *ctx) {
    if (!ctx->prof_pool.empty()) {
        hipEvent_t e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void pmf_prof_begin_on(pmf_ctx *ctx, int kernel, hipStream_t stream) {
    if (!ctx->prof) return;
    pmf_ctx::ProfRec r;
    r.a = take_event(ctx);
    r.b = take_event(ctx);
    r.kernel = kernel;
    (void)hipEventRecord(r.a, stream);
    ctx->prof_pending.push_back(r);
}

void pmf_prof_end_on(pmf_ctx *ctx, hipStream_t stream) {
    if (!ctx->prof || ctx->prof_pending.empty()) return;
    (void)hipEventRecord(ctx->prof_pending.back().b, stream);
}

void pmf_prof_begin(pmf_ctx *ctx, int kernel) { pmf_prof_begin_on(ctx, kernel, ctx->stream); }
void pmf_prof_end(pmf_ctx *ctx) { pmf_prof_end_on(ctx, ctx->stream); }

static int prof_drain(pmf_ctx *ctx) {
    if (ctx->prof_pending.empty()) return PMF_OK;
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof_pending) {
        float ms = 0.f;
        PMF_HIP_CHECK(hipEventSynchronize(r.b));   // brackets on the collective stream end there
        PMF_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
        ctx->prof_ms[r.kernel] += ms;
        ctx->prof_n[r.kernel] += 1;
        ctx->prof_pool.push_back(r.a);
        ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_pending.clear();
    return PMF_OK;
}

extern "C" int pmf_prof_enable(pmf_ctx *ctx, int enable) {
    CHECK_CTX(ctx, "pmf_prof_enable");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = prof_drain(ctx);
    ctx->prof = enable != 0;
    return rc;
}

extern "C" int pmf_prof_reset(pmf_ctx *ctx) {
    CHECK_CTX(ctx, "pmf_prof_reset");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = prof_drain(ctx);
    for (int k = 0; k < PMF_KERNEL_COUNT; ++k) {
        ctx->prof_ms[k] = 0;
        ctx->prof_n[k] = 0;
    }
    return rc;
}

extern "C" int pmf_prof_get(pmf_ctx *ctx, int kernel, double *total_ms, int64_t *launches) {
    CHECK_CTX(ctx, "pmf_prof_get");
    PMF_REQUIRE(kernel >= 0 && kernel < PMF_KERNEL_COUNT, PMF_EINVAL, "pmf_prof_get: bad kernel id %d", kernel);
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    int rc = prof_drain(ctx);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->prof_ms[kernel];
    if (launches) *launches = ctx->prof_n[kernel];
    return PMF_OK;
}
