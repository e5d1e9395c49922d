// Gaussian MF, MAP estimation by stochastic gradient steps -- SURVEY.md section 8(f) rank 4.
//
// The reference has NO gradient loop for the Gaussian model: this mode has no
// reference counterpart and its parity is UNPINNED.  It optimises the MAP objective of the model of
// gaussian_mf_cavi_bias.py (x_ij ~ N(b_i + b_j + theta_i . beta_j, sigma2), N(0, eta2 I) priors),
//
//   L = 1/(2 sigma2) sum (x - b_i - b_j - theta_i . beta_j)^2 + |theta|^2/(2 eta_theta2) + ...
//
// with per-rating steps (one K-length dot, one K-length AXPY):
//
//   e      = x - b_r - b_o - theta_r . beta_o
//   theta_r += lr ( e beta_o / sigma2 - theta_r / (eta2 n_r) )
//   b_r     += lr ( e / sigma2        - b_r     / (eta_b2 n_r) )
//
// (the prior is spread over the row's n_r ratings, the weighting hpf_pytorch.py:71-184 uses for its
// priors).  To stay deterministic and atomic-free the epoch alternates like the CAVI sweeps: all
// rows of one side walk through their own ratings in input order with the other side fixed, rows
// independent of each other.  A row cut into several tasks (more than 256 ratings) -- and, on
// several GPUs, an item whose ratings live on several ranks -- runs every piece from the row's old
// value and takes the rating-count-weighted average of the pieces' displacements ("parallelized
// SGD" model averaging): the statistics are [rows x (Kpad + 4)] = sum len * d_theta | sum len * d_b
// | sum len, additive across tasks and ranks.
#include <type_traits>

#include "pmf_device.h"
#include "pmf_internal.h"

namespace {

template <typename T>
struct SgdParams {
    const PmfTask *tasks;
    int64_t n_tasks;
    const PmfSplitRow *split;
    int64_t n_split;
    const int64_t *ptr;
    const int32_t *other;
    const T *val;
    T *factor_self;
    const T *factor_other;
    T *bias_self;          // null: model without biases
    const T *bias_other;
    T *partial;            // [n_slots][width]
    T *stats;              // [rows][width]
    T lr, inv_sigma2, inv_eta2, inv_eta_bias2;
    int kpad, width;
    int64_t row0, row1;
};

// lane group per task, sequential over the task's ratings
template <typename T, int LPR>
__global__ __launch_bounds__(256) void gauss_sgd_kernel(SgdParams<T> p) {
    constexpr int G = 256 / LPR;
    constexpr int UN = LPR < 8 ? LPR : 8;   // gathers in flight per lane (independent of the running row; 16 measured the same)
    const int c = threadIdx.x % LPR;
    const int64_t task_id = (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    if (task_id >= p.n_tasks) return;
    const PmfTask t = p.tasks[task_id];
    const int koff = c * PMF_VEC;
    const bool active = koff < p.kpad;
    const bool has_bias = p.bias_self != nullptr;
    const T n_row = (T)(p.ptr[t.row + 1] - p.ptr[t.row]);
    const T shrink_t = p.lr * p.inv_eta2 / n_row, shrink_b = p.lr * p.inv_eta_bias2 / n_row;
    const T step = p.lr * p.inv_sigma2;
    const Vec4<T> start = active ? load4(p.factor_self + (int64_t)t.row * p.kpad + koff) : zero4<T>();
    const T b_start = has_bias ? p.bias_self[t.row] : (T)0;
    Vec4<T> th = start;
    T b = b_start;
    const int32_t *col = p.other + t.start;
    const T *val = p.val + t.start;
    for (int base = 0; base < t.len; base += LPR) {
        const int n = min(LPR, t.len - base);
        int my_o = 0;
        T my_x = (T)0;
        if (c < n) {
            my_o = col[base + c];
            my_x = val[base + c] - (has_bias ? p.bias_other[my_o] : (T)0);
        }
        for (int tt = 0; tt < n; tt += UN) {
            int o[UN];
            T xv[UN];
            Vec4<T> be[UN];
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                o[q] = __shfl(my_o, tt + q, LPR);
                xv[q] = __shfl(my_x, tt + q, LPR);
            }
#pragma unroll
            for (int q = 0; q < UN; ++q)   // the gathers do not depend on theta: UN of them in flight
                be[q] = active ? load4(p.factor_other + (int64_t)o[q] * p.kpad + koff) : zero4<T>();
#pragma unroll
            for (int q = 0; q < UN; ++q) {
                if (tt + q < n) {
                    T d = be[q].v[0] * th.v[0];
                    d = fma(be[q].v[1], th.v[1], d);
                    d = fma(be[q].v[2], th.v[2], d);
                    d = fma(be[q].v[3], th.v[3], d);
                    d = group_sum<LPR>(d);
                    const T e = xv[q] - b - d;
#pragma unroll
                    for (int k = 0; k < PMF_VEC; ++k) th.v[k] = fma(step * e, be[q].v[k], th.v[k] - shrink_t * th.v[k]);
                    if (has_bias) b = fma(step, e, b - shrink_b * b);
                }
            }
        }
    }
    // len-weighted displacement of this piece
    const T w = (T)t.len;
    T *dst = (t.slot >= 0 ? p.partial + (int64_t)t.slot * p.width : p.stats + (int64_t)t.row * p.width);
    if (active) {
        Vec4<T> d;
#pragma unroll
        for (int k = 0; k < PMF_VEC; ++k) d.v[k] = w * (th.v[k] - start.v[k]);
        store4(dst + koff, d);
    }
    if (c == 0) {
        Vec4<T> tail = zero4<T>();
        tail.v[0] = w * (b - b_start);
        tail.v[1] = w;
        store4(dst + p.kpad, tail);
    }
}

// split rows: the slots of a row are summed by G slot lanes per element (slot q goes to lane
// q mod G), the G partial sums are then added in lane order -- a fixed order, so deterministic.
// (A popular item has thousands of slots: one thread per element walking all of them took 1.5 ms.)
template <typename T>
__global__ __launch_bounds__(1024) void gauss_sgd_combine_kernel(SgdParams<T> p, int lanes_k) {
    extern __shared__ __align__(16) unsigned char sgd_smem[];
    T *red = reinterpret_cast<T *>(sgd_smem);   // [G][lanes_k]
    const int64_t s = blockIdx.x;
    const PmfSplitRow sr = p.split[s];
    const int G = blockDim.x / lanes_k;
    const int k = threadIdx.x % lanes_k, g = threadIdx.x / lanes_k;
    T sum = (T)0;
    if (k < p.width)
        for (int q = g; q < sr.n_slots; q += G) sum += p.partial[(int64_t)(sr.first_slot + q) * p.width + k];
    red[g * lanes_k + k] = sum;
    __syncthreads();
    if (g == 0 && k < p.width) {
        T total = (T)0;
        for (int gg = 0; gg < G; ++gg) total += red[gg * lanes_k + k];
        p.stats[(int64_t)sr.row * p.width + k] = total;
    }
}

// value += weighted displacement / weight, for every row of [row0, row1) that has ratings anywhere
template <typename T>
__global__ void gauss_sgd_finalize_kernel(SgdParams<T> p) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = p.kpad + 1;
    const int64_t row = p.row0 + idx / per;
    const int k = (int)(idx % per);
    if (row >= p.row1) return;
    const T *st = p.stats + row * p.width;
    const T cnt = st[p.kpad + 1];
    if (!(cnt > (T)0)) return;
    if (k < p.kpad) p.factor_self[row * p.kpad + k] += st[k] / cnt;
    else if (p.bias_self) p.bias_self[row] += st[p.kpad] / cnt;
}

template <typename T, int LPR>
void launch_sgd(pmf_ctx *ctx, const SgdParams<T> &p) {
    constexpr int G = 256 / LPR;
    hipLaunchKernelGGL((gauss_sgd_kernel<T, LPR>), dim3((unsigned)((p.n_tasks + G - 1) / G)), dim3(256), 0, ctx->stream, p);
}

// mode 1: accumulate into stats; mode 2: finalize from stats
template <typename T>
int run_sgd(pmf_ctx *ctx, int side, int mode, void *stats, double lr, double sigma2, double eta2, double eta_bias2) {
    const int other = 1 - side;
    const PmfSideIndex &ix = ctx->index[side];
    PMF_REQUIRE(ix.d_ptr, PMF_EINVAL, "pmf_gauss_sgd_sweep: ratings have not been set");
    int rc;
    if ((rc = pmf_require_array(ctx, side, PMF_ARR_FACTOR, "pmf_gauss_sgd_sweep"))) return rc;
    if ((rc = pmf_require_array(ctx, other, PMF_ARR_FACTOR, "pmf_gauss_sgd_sweep"))) return rc;
    const bool bias = ctx->arr[0][PMF_ARR_BIAS] != nullptr && ctx->arr[1][PMF_ARR_BIAS] != nullptr;
    const PmfTaskView tl = pmf_task_view(ctx, side, ix.sgd_tasks, true);
    SgdParams<T> p;
    p.tasks = tl.d_tasks;
    p.n_tasks = tl.n_tasks;
    p.split = tl.d_split;
    p.n_split = tl.n_split;
    p.ptr = ix.d_ptr;
    p.other = ix.d_other;
    p.val = (const T *)ix.d_val;
    p.factor_self = (T *)ctx->arr[side][PMF_ARR_FACTOR];
    p.factor_other = (const T *)ctx->arr[other][PMF_ARR_FACTOR];
    p.bias_self = bias ? (T *)ctx->arr[side][PMF_ARR_BIAS] : nullptr;
    p.bias_other = bias ? (const T *)ctx->arr[other][PMF_ARR_BIAS] : nullptr;
    p.kpad = ctx->kpad;
    p.width = ctx->kpad + PMF_VEC;
    p.stats = (T *)stats;
    p.row0 = tl.row0;
    p.row1 = tl.row1;
    if (mode == 1) {
        PMF_REQUIRE(lr > 0 && sigma2 > 0 && eta2 > 0 && eta_bias2 > 0, PMF_EINVAL,
                    "pmf_gauss_sgd_sweep: lr and the variances must be positive");
        if (tl.n_slots > 0)
            if ((rc = pmf_ensure_partial(ctx, (size_t)tl.n_slots * p.width * sizeof(T)))) return rc;
        p.partial = (T *)ctx->d_partial;
        p.lr = (T)lr;
        p.inv_sigma2 = (T)(1.0 / sigma2);
        p.inv_eta2 = (T)(1.0 / eta2);
        p.inv_eta_bias2 = (T)(1.0 / eta_bias2);
        if (tl.row1 > tl.row0)  // rows without ratings on this rank contribute zeros
            PMF_HIP_CHECK(hipMemsetAsync((T *)stats + tl.row0 * p.width, 0, (size_t)(tl.row1 - tl.row0) * p.width * sizeof(T),
                                         ctx->stream));
        PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_SGD);
        if (tl.n_tasks > 0) {
            switch (pmf_lanes_per_row(ctx->kpad)) {
                case 1: launch_sgd<T, 1>(ctx, p); break;
                case 2: launch_sgd<T, 2>(ctx, p); break;
                case 4: launch_sgd<T, 4>(ctx, p); break;
                case 8: launch_sgd<T, 8>(ctx, p); break;
                case 16: launch_sgd<T, 16>(ctx, p); break;
                case 32: launch_sgd<T, 32>(ctx, p); break;
                default: launch_sgd<T, 64>(ctx, p); break;
            }
        }
        if (tl.n_split > 0) {
            int lanes_k = 1;
            while (lanes_k < p.width) lanes_k <<= 1;   // width <= 260 -> at most 512
            hipLaunchKernelGGL((gauss_sgd_combine_kernel<T>), dim3((unsigned)tl.n_split), dim3(1024),
                               (size_t)1024 * sizeof(T), ctx->stream, p, lanes_k);
        }
    } else {
        PmfProfScope prof(ctx, PMF_KERNEL_GAUSS_SGD);
        const int64_t n = (tl.row1 - tl.row0) * (int64_t)(ctx->kpad + 1);
        if (n > 0)
            hipLaunchKernelGGL((gauss_sgd_finalize_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, p);
    }
    PMF_HIP_CHECK(hipGetLastError());
    return PMF_OK;
}

}  // namespace

#define SGD_PROLOGUE(fn)                                                                                \
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, fn ": null context");                                       \
    PMF_REQUIRE(side == PMF_SIDE_USER || side == PMF_SIDE_ITEM, PMF_EINVAL, fn ": bad side %d", side);  \
    PMF_HIP_CHECK(hipSetDevice(ctx->device));

extern "C" int pmf_ctx_sgd_stats_width(pmf_ctx *ctx, int *width) {
    PMF_REQUIRE(ctx != nullptr && width != nullptr, PMF_EINVAL, "pmf_ctx_sgd_stats_width: null argument");
    *width = ctx->kpad + PMF_VEC;
    return PMF_OK;
}

extern "C" int pmf_gauss_sgd_accumulate(pmf_ctx *ctx, int side, void *stats_dev, double lr, double sigma2, double eta2,
                                        double eta_bias2) {
    SGD_PROLOGUE("pmf_gauss_sgd_accumulate");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_sgd_accumulate: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_sgd<double>(ctx, side, 1, stats_dev, lr, sigma2, eta2, eta_bias2);
    return run_sgd<float>(ctx, side, 1, stats_dev, lr, sigma2, eta2, eta_bias2);
}

extern "C" int pmf_gauss_sgd_finalize(pmf_ctx *ctx, int side, const void *stats_dev) {
    SGD_PROLOGUE("pmf_gauss_sgd_finalize");
    PMF_REQUIRE(stats_dev, PMF_EINVAL, "pmf_gauss_sgd_finalize: null stats buffer");
    if (ctx->dtype == PMF_F64) return run_sgd<double>(ctx, side, 2, (void *)stats_dev, 1, 1, 1, 1);
    return run_sgd<float>(ctx, side, 2, (void *)stats_dev, 1, 1, 1, 1);
}

extern "C" int pmf_gauss_sgd_sweep(pmf_ctx *ctx, int side, double lr, double sigma2, double eta2, double eta_bias2) {
    SGD_PROLOGUE("pmf_gauss_sgd_sweep");
    if (side == PMF_SIDE_ITEM && pmf_comm_active(ctx)) {
        // several ranks: the items' rating-count-weighted displacement sums are all-reduced (pmf_comm.hip)
        const size_t width = (size_t)ctx->kpad + PMF_VEC;
        void *stats = nullptr;
        int rc = pmf_comm_stats(ctx, 0, (size_t)ctx->rows[side] * width * ctx->elem, &stats);
        if (rc) return rc;
        PmfExchange ex;
        ex.arrays[ex.n_arrays++] = PMF_ARR_FACTOR;
        if (ctx->arr[0][PMF_ARR_BIAS] != nullptr && ctx->arr[1][PMF_ARR_BIAS] != nullptr) ex.arrays[ex.n_arrays++] = PMF_ARR_BIAS;
        return pmf_comm_half_sweep(
            ctx, side, width, stats, true,
            [&] { return pmf_gauss_sgd_accumulate(ctx, side, stats, lr, sigma2, eta2, eta_bias2); },
            [&] { return pmf_gauss_sgd_finalize(ctx, side, stats); }, ex);
    }
    const int saved = ctx->cur_chunk[side];
    ctx->cur_chunk[side] = -1;  // the one-call form always covers every row
    const size_t bytes = (size_t)ctx->rows[side] * (ctx->kpad + PMF_VEC) * ctx->elem;
    int rc = pmf_ensure_scratch(ctx, bytes);
    if (!rc) rc = pmf_gauss_sgd_accumulate(ctx, side, ctx->d_scratch, lr, sigma2, eta2, eta_bias2);
    if (!rc) rc = pmf_gauss_sgd_finalize(ctx, side, ctx->d_scratch);
    ctx->cur_chunk[side] = saved;
    return rc;
}
