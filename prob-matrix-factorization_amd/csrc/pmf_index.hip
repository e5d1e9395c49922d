// Device-side build of the two rating orders (CSR by user, CSC by item).
//
// Replaces `_build_index_lists` (hpf_cavi.py:97-107, poisson_mf_cavi.py:73-84,
// gaussian_mf_cavi_bias.py:69-86): the reference appends every rating to the list
// of its user and of its item in input order.  Here the (user, item, rating)
// triples are uploaded once and, per side, a STABLE radix sort of the positions
// by row id (rocPRIM, least-significant-digit passes) gives the same within-row
// order; the opposite-side ids and the ratings are then gathered into row order
// and the row pointers are found by binary search in the sorted keys.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <limits.h>

#include "pmf_internal.h"

namespace {

constexpr unsigned long long NO_BAD = ~0ull;

__global__ void check_ids_kernel(const int32_t *u, const int32_t *i, int64_t nnz, int64_t U, int64_t I,
                                 unsigned long long *first_bad) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= nnz) return;
    const int32_t a = u[n], b = i[n];
    if (a < 0 || a >= U || b < 0 || b >= I) atomicMin(first_bad, (unsigned long long)n);
}

__global__ void iota_kernel(uint32_t *p, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) p[k] = (uint32_t)k;
}

template <typename T>
__global__ void gather_kernel(const uint32_t *perm, const int32_t *other_in, const double *x_in, int32_t *other_out,
                              T *val_out, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t src = perm[k];
    other_out[k] = other_in[src];
    val_out[k] = (T)x_in[src];
}

// ptr[r] = number of ratings whose row id is < r (r = 0 .. rows)
__global__ void row_ptr_kernel(const uint32_t *sorted_keys, int64_t nnz, int64_t rows, int64_t *ptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted_keys[mid] < r) lo = mid + 1;
        else hi = mid;
    }
    ptr[r] = lo;
}

struct Scratch {  // temporaries of one build; freed on every exit path
    std::vector<void *> p;
    ~Scratch() {
        for (void *q : p) (void)hipFree(q);
    }
    template <typename T>
    hipError_t alloc(T **out, size_t bytes) {
        void *q = nullptr;
        hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
        if (e == hipSuccess) p.push_back(q);
        *out = (T *)q;
        return e;
    }
};

inline unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

inline unsigned key_bits(int64_t rows) {  // row ids are < rows
    unsigned bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < rows) ++bits;
    return bits;
}

}  // namespace

// Upload + validate.  On success the device copies live in `s` and *bad < 0; a bad id gives its
// position in *bad (no device state of the context has been touched yet).
static int upload_and_check(pmf_ctx *ctx, Scratch &s, int64_t nnz, const int32_t *user_ids, const int32_t *item_ids,
                            const double *ratings, int32_t **d_u, int32_t **d_i, double **d_x, int64_t *bad) {
    *bad = -1;
    PMF_HIP_CHECK(s.alloc(d_u, (size_t)nnz * sizeof(int32_t)));
    PMF_HIP_CHECK(s.alloc(d_i, (size_t)nnz * sizeof(int32_t)));
    PMF_HIP_CHECK(s.alloc(d_x, (size_t)nnz * sizeof(double)));
    unsigned long long *d_bad = nullptr;
    PMF_HIP_CHECK(s.alloc(&d_bad, sizeof(unsigned long long)));
    if (nnz == 0) return PMF_OK;
    PMF_HIP_CHECK(hipMemcpyAsync(*d_u, user_ids, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PMF_HIP_CHECK(hipMemcpyAsync(*d_i, item_ids, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    PMF_HIP_CHECK(hipMemcpyAsync(*d_x, ratings, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    PMF_HIP_CHECK(hipMemsetAsync(d_bad, 0xff, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(check_ids_kernel, dim3(grid_for(nnz)), dim3(256), 0, ctx->stream, *d_u, *d_i, nnz, ctx->rows[0],
                       ctx->rows[1], d_bad);
    PMF_HIP_CHECK(hipGetLastError());
    unsigned long long h_bad = NO_BAD;
    PMF_HIP_CHECK(hipMemcpyAsync(&h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost, ctx->stream));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h_bad != NO_BAD) *bad = (int64_t)h_bad;
    return PMF_OK;
}

template <typename T>
static int order_side(pmf_ctx *ctx, Scratch &s, int side, int64_t nnz, const int32_t *d_key, const int32_t *d_oth,
                      const double *d_x, uint32_t *d_keys_out, uint32_t *d_pos_in, uint32_t *d_pos_out, void *d_tmp,
                      size_t tmp_bytes) {
    PmfSideIndex &ix = ctx->index[side];
    const int64_t rows = ctx->rows[side];
    if (nnz > 0) {
        const unsigned bits = key_bits(rows);
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nnz)), dim3(256), 0, ctx->stream, d_pos_in, nnz);
        size_t bytes = tmp_bytes;
        PMF_HIP_CHECK(rocprim::radix_sort_pairs(d_tmp, bytes, (const uint32_t *)d_key, d_keys_out, (const uint32_t *)d_pos_in,
                                                d_pos_out, (size_t)nnz, 0u, bits, ctx->stream));
        hipLaunchKernelGGL((gather_kernel<T>), dim3(grid_for(nnz)), dim3(256), 0, ctx->stream, d_pos_out, d_oth, d_x,
                           ix.d_other, (T *)ix.d_val, nnz);
    }
    hipLaunchKernelGGL(row_ptr_kernel, dim3(grid_for(rows + 1)), dim3(256), 0, ctx->stream, d_keys_out, nnz, rows, ix.d_ptr);
    PMF_HIP_CHECK(hipGetLastError());
    ix.h_ptr.resize((size_t)rows + 1);
    PMF_HIP_CHECK(hipMemcpyAsync(ix.h_ptr.data(), ix.d_ptr, (size_t)(rows + 1) * sizeof(int64_t), hipMemcpyDeviceToHost,
                                 ctx->stream));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    (void)s;
    return PMF_OK;
}

// Two-phase interface used by pmf_ctx_set_ratings (pmf_ctx.hip):
//   pmf_index_device_begin  uploads and validates (context untouched on failure),
//   pmf_index_device_finish fills index[side].{d_ptr, d_other, d_val, h_ptr} (already allocated).
struct PmfIndexBuild {
    Scratch s;
    int32_t *d_u = nullptr, *d_i = nullptr;
    double *d_x = nullptr;
};

int pmf_index_device_begin(pmf_ctx *ctx, int64_t nnz, const int32_t *user_ids, const int32_t *item_ids,
                           const double *ratings, PmfIndexBuild **out, int64_t *bad_position) {
    PmfIndexBuild *b = new PmfIndexBuild();
    int rc = upload_and_check(ctx, b->s, nnz, user_ids, item_ids, ratings, &b->d_u, &b->d_i, &b->d_x, bad_position);
    if (rc != PMF_OK || *bad_position >= 0) {
        delete b;
        b = nullptr;
    }
    *out = b;
    return rc;
}

void pmf_index_device_abort(PmfIndexBuild *b) { delete b; }

int pmf_index_device_finish(pmf_ctx *ctx, PmfIndexBuild *b, int64_t nnz) {
    struct Guard {
        PmfIndexBuild *b;
        ~Guard() { delete b; }
    } guard{b};
    uint32_t *keys_out = nullptr, *pos_in = nullptr, *pos_out = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    PMF_HIP_CHECK(b->s.alloc(&keys_out, (size_t)nnz * sizeof(uint32_t)));
    PMF_HIP_CHECK(b->s.alloc(&pos_in, (size_t)nnz * sizeof(uint32_t)));
    PMF_HIP_CHECK(b->s.alloc(&pos_out, (size_t)nnz * sizeof(uint32_t)));
    if (nnz > 0) {
        for (int side = 0; side < 2; ++side) {  // the larger of the two sorts' temporary storage
            size_t need = 0;
            PMF_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, need, (const uint32_t *)b->d_u, keys_out,
                                                    (const uint32_t *)pos_in, pos_out, (size_t)nnz, 0u,
                                                    key_bits(ctx->rows[side]), ctx->stream));
            if (need > tmp_bytes) tmp_bytes = need;
        }
        PMF_HIP_CHECK(b->s.alloc(&tmp, tmp_bytes));
    }
    for (int side = 0; side < 2; ++side) {
        const int32_t *key = side == PMF_SIDE_USER ? b->d_u : b->d_i;
        const int32_t *oth = side == PMF_SIDE_USER ? b->d_i : b->d_u;
        int rc = ctx->dtype == PMF_F64
                     ? order_side<double>(ctx, b->s, side, nnz, key, oth, b->d_x, keys_out, pos_in, pos_out, tmp, tmp_bytes)
                     : order_side<float>(ctx, b->s, side, nnz, key, oth, b->d_x, keys_out, pos_in, pos_out, tmp, tmp_bytes);
        if (rc) return rc;
    }
    return PMF_OK;
}
