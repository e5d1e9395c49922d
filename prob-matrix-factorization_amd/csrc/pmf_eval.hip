// predict / evaluate / top-k kernels (rows a10, a11 of SURVEY.md section 8 and
// the "identical top-k" check of the north star).
#include <algorithm>
#include <vector>

#include "pmf_device.h"

template <typename T>
struct PredictParams {
    const int32_t *u;
    const int32_t *i;
    int64_t n;
    const T *fu;
    const T *fi;
    const T *bu;  // null when biases are not used
    const T *bi;
    const T *su;  // null unless the scalar factors of the extended Poisson model apply
    const T *si;
    int64_t n_users, n_items;
    int kpad;
    double offset;
};

// dot(FACTOR_user[u], FACTOR_item[i]) (+ biases) for one pair, computed by a
// lane group; ids outside the trained dimensions give 0 (hpf_cavi.py:220-229).
template <typename T, int LPR>
__device__ __forceinline__ double predict_pair(const PredictParams<T> &p, int64_t idx, int c) {
    const int u = p.u[idx], i = p.i[idx];
    const bool valid = u >= 0 && i >= 0 && u < p.n_users && i < p.n_items;
    const int koff = c * PMF_VEC;
    T d = (T)0;
    if (valid && koff < p.kpad) {
        Vec4<T> a = load4(p.fu + (int64_t)u * p.kpad + koff);
        Vec4<T> b = load4(p.fi + (int64_t)i * p.kpad + koff);
        d = a.v[0] * b.v[0];
        d = fma(a.v[1], b.v[1], d);
        d = fma(a.v[2], b.v[2], d);
        d = fma(a.v[3], b.v[3], d);
    }
    d = group_sum<LPR>(d);
    if (valid && p.su) d = p.su[u] * p.si[i] * d;
    if (valid && p.bu) d = p.bu[u] + p.bi[i] + d;
    return (valid ? (double)d : 0.0) + p.offset;
}

template <typename T, int LPR>
__global__ __launch_bounds__(256) void predict_kernel(PredictParams<T> p, double *out) {
    constexpr int G = 256 / LPR;
    const int c = threadIdx.x % LPR;
    const int64_t stride = (int64_t)gridDim.x * G;
    // all groups of a wavefront iterate the same number of times (DPP inside)
    const int64_t rounds = (p.n + stride - 1) / stride;
    int64_t idx = (int64_t)blockIdx.x * G + threadIdx.x / LPR;
    for (int64_t r = 0; r < rounds; ++r, idx += stride) {
        if (idx < p.n) {
            double v = predict_pair<T, LPR>(p, idx, c);
            if (c == 0) out[idx] = v;
        }
    }
}

// Fused validation monitor: per-block partial sums of squared error and of the
// per-label absolute error / count, combined in block order on the host
// (deterministic; metrics.py:6-10, :37-51).
template <typename T, int LPR>
__global__ __launch_bounds__(256) void eval_kernel(PredictParams<T> p, const double *y, const int32_t *label,
                                                   int n_labels, double *block_out) {
    constexpr int G = 256 / LPR;
    __shared__ double s_sse[G];
    __shared__ double s_abs[G][PMF_MAX_LABELS];
    __shared__ int s_cnt[G][PMF_MAX_LABELS];
    const int c = threadIdx.x % LPR;
    const int g = threadIdx.x / LPR;
    for (int t = threadIdx.x; t < G * PMF_MAX_LABELS; t += 256) {
        (&s_abs[0][0])[t] = 0.0;
        (&s_cnt[0][0])[t] = 0;
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * G;
    const int64_t rounds = (p.n + stride - 1) / stride;
    int64_t idx = (int64_t)blockIdx.x * G + g;
    double sse = 0.0;
    for (int64_t r = 0; r < rounds; ++r, idx += stride) {
        if (idx < p.n) {
            double err = y[idx] - predict_pair<T, LPR>(p, idx, c);
            if (c == 0) {
                sse += err * err;
                int l = label[idx];
                s_abs[g][l] += fabs(err);
                s_cnt[g][l] += 1;
            }
        }
    }
    if (c == 0) s_sse[g] = sse;
    __syncthreads();
    double *dst = block_out + (int64_t)blockIdx.x * (1 + 2 * PMF_MAX_LABELS);
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int k = 0; k < G; ++k) s += s_sse[k];
        dst[0] = s;
    }
    if (threadIdx.x < n_labels) {
        double a = 0.0;
        long long n = 0;
        for (int k = 0; k < G; ++k) {
            a += s_abs[k][threadIdx.x];
            n += s_cnt[k][threadIdx.x];
        }
        dst[1 + threadIdx.x] = a;
        dst[1 + PMF_MAX_LABELS + threadIdx.x] = (double)n;
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T>
static int fill_params(pmf_ctx *ctx, int use_bias, double offset, PredictParams<T> &p, const char *fn) {
    int rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_FACTOR, fn))) return rc;
    if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_FACTOR, fn))) return rc;
    const bool scale = (use_bias & PMF_PREDICT_SCALE) != 0;
    use_bias &= PMF_PREDICT_BIAS;
    if (use_bias) {
        if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_BIAS, fn))) return rc;
        if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_BIAS, fn))) return rc;
    }
    if (scale) {
        if ((rc = pmf_require_array(ctx, PMF_SIDE_USER, PMF_ARR_SCALE, fn))) return rc;
        if ((rc = pmf_require_array(ctx, PMF_SIDE_ITEM, PMF_ARR_SCALE, fn))) return rc;
    }
    p.su = scale ? (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_SCALE] : nullptr;
    p.si = scale ? (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_SCALE] : nullptr;
    p.fu = (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_FACTOR];
    p.fi = (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_FACTOR];
    p.bu = use_bias ? (const T *)ctx->arr[PMF_SIDE_USER][PMF_ARR_BIAS] : nullptr;
    p.bi = use_bias ? (const T *)ctx->arr[PMF_SIDE_ITEM][PMF_ARR_BIAS] : nullptr;
    p.n_users = ctx->rows[0];
    p.n_items = ctx->rows[1];
    p.kpad = ctx->kpad;
    p.offset = offset;
    return PMF_OK;
}

static inline int eval_lpr(const pmf_ctx *ctx) { return std::max(4, pmf_lanes_per_row(ctx->kpad)); }

template <typename T, int LPR>
static void launch_predict(pmf_ctx *ctx, const PredictParams<T> &p, double *out) {
    constexpr int G = 256 / LPR;
    int grid = (int)std::min<int64_t>((p.n + G - 1) / G, 8192);
    hipLaunchKernelGGL((predict_kernel<T, LPR>), dim3(grid), dim3(256), 0, ctx->stream, p, out);
}

template <typename T>
static int run_predict(pmf_ctx *ctx, int64_t n, const int32_t *u, const int32_t *i, int use_bias,
                       double offset, double *out) {
    PredictParams<T> p;
    int rc = fill_params(ctx, use_bias, offset, p, "pmf_predict");
    if (rc) return rc;
    const int64_t step = 4 << 20;  // pairs per staging round
    const int64_t m = std::min(n, step);
    if ((rc = pmf_ensure_scratch(ctx, (size_t)m * 16))) return rc;
    int32_t *d_u = (int32_t *)ctx->d_scratch;
    int32_t *d_i = d_u + m;
    double *d_out = (double *)(d_i + m);
    for (int64_t at = 0; at < n; at += step) {
        const int64_t cnt = std::min(step, n - at);
        PMF_HIP_CHECK(hipMemcpyAsync(d_u, u + at, (size_t)cnt * 4, hipMemcpyHostToDevice, ctx->stream));
        PMF_HIP_CHECK(hipMemcpyAsync(d_i, i + at, (size_t)cnt * 4, hipMemcpyHostToDevice, ctx->stream));
        p.u = d_u;
        p.i = d_i;
        p.n = cnt;
        {
            PmfProfScope prof(ctx, PMF_KERNEL_PREDICT);
            switch (eval_lpr(ctx)) {
                case 4: launch_predict<T, 4>(ctx, p, d_out); break;
                case 8: launch_predict<T, 8>(ctx, p, d_out); break;
                case 16: launch_predict<T, 16>(ctx, p, d_out); break;
                case 32: launch_predict<T, 32>(ctx, p, d_out); break;
                default: launch_predict<T, 64>(ctx, p, d_out); break;
            }
        }
        PMF_HIP_CHECK(hipGetLastError());
        PMF_HIP_CHECK(hipMemcpyAsync(out + at, d_out, (size_t)cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
        PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    return PMF_OK;
}

extern "C" int pmf_predict(pmf_ctx *ctx, int64_t n, const int32_t *user_ids, const int32_t *item_ids,
                           int use_bias, double offset, double *out) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_predict: null context");
    PMF_REQUIRE(n >= 0, PMF_EINVAL, "pmf_predict: negative n");
    if (n == 0) return PMF_OK;
    PMF_REQUIRE(user_ids && item_ids && out, PMF_EINVAL, "pmf_predict: null argument");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->dtype == PMF_F64) return run_predict<double>(ctx, n, user_ids, item_ids, use_bias, offset, out);
    return run_predict<float>(ctx, n, user_ids, item_ids, use_bias, offset, out);
}

extern "C" int pmf_eval_set(pmf_ctx *ctx, int64_t n, const int32_t *user_ids, const int32_t *item_ids,
                            const double *y_true, const int32_t *label_index, int n_labels) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_eval_set: null context");
    PMF_REQUIRE(n > 0 && user_ids && item_ids && y_true && label_index, PMF_EINVAL,
                "pmf_eval_set: empty or null input");
    PMF_REQUIRE(n_labels >= 1 && n_labels <= PMF_MAX_LABELS, PMF_ERANGE,
                "pmf_eval_set: n_labels=%d outside [1, %d]", n_labels, PMF_MAX_LABELS);
    for (int64_t k = 0; k < n; ++k)
        PMF_REQUIRE(label_index[k] >= 0 && label_index[k] < n_labels, PMF_ERANGE,
                    "pmf_eval_set: label index %d at position %lld outside [0, %d)", label_index[k],
                    (long long)k, n_labels);
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    PmfEvalSet &ev = ctx->eval;
    pmf_dev_free(ctx, ev.d_u, (size_t)ev.n * 4);
    pmf_dev_free(ctx, ev.d_i, (size_t)ev.n * 4);
    pmf_dev_free(ctx, ev.d_y, (size_t)ev.n * 8);
    pmf_dev_free(ctx, ev.d_label, (size_t)ev.n * 4);
    ev = PmfEvalSet();
    int rc;
    if ((rc = pmf_dev_alloc(ctx, (void **)&ev.d_u, (size_t)n * 4))) return rc;
    if ((rc = pmf_dev_alloc(ctx, (void **)&ev.d_i, (size_t)n * 4))) return rc;
    if ((rc = pmf_dev_alloc(ctx, (void **)&ev.d_y, (size_t)n * 8))) return rc;
    if ((rc = pmf_dev_alloc(ctx, (void **)&ev.d_label, (size_t)n * 4))) return rc;
    ev.n = n;
    ev.n_labels = n_labels;
    PMF_HIP_CHECK(hipMemcpy(ev.d_u, user_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    PMF_HIP_CHECK(hipMemcpy(ev.d_i, item_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    PMF_HIP_CHECK(hipMemcpy(ev.d_y, y_true, (size_t)n * 8, hipMemcpyHostToDevice));
    PMF_HIP_CHECK(hipMemcpy(ev.d_label, label_index, (size_t)n * 4, hipMemcpyHostToDevice));
    return PMF_OK;
}

template <typename T, int LPR>
static void launch_eval(pmf_ctx *ctx, const PredictParams<T> &p, int grid, double *block_out) {
    hipLaunchKernelGGL((eval_kernel<T, LPR>), dim3(grid), dim3(256), 0, ctx->stream, p, ctx->eval.d_y,
                       ctx->eval.d_label, ctx->eval.n_labels, block_out);
}

template <typename T>
static int run_eval(pmf_ctx *ctx, int use_bias, double offset, double *sse, double *abs_l, int64_t *cnt_l) {
    PredictParams<T> p;
    int rc = fill_params(ctx, use_bias, offset, p, "pmf_eval_run");
    if (rc) return rc;
    const PmfEvalSet &ev = ctx->eval;
    p.u = ev.d_u;
    p.i = ev.d_i;
    p.n = ev.n;
    const int lpr = eval_lpr(ctx);
    const int G = 256 / lpr;
    const int grid = (int)std::min<int64_t>((ev.n + G - 1) / G, 1024);
    const size_t rec = 1 + 2 * PMF_MAX_LABELS;
    const size_t bytes = (size_t)grid * rec * sizeof(double);
    if ((rc = pmf_ensure_scratch(ctx, bytes))) return rc;
    if ((rc = pmf_ensure_pinned(ctx, bytes))) return rc;
    double *block_out = (double *)ctx->d_scratch;
    {
        PmfProfScope prof(ctx, PMF_KERNEL_EVAL);
        switch (lpr) {
            case 4: launch_eval<T, 4>(ctx, p, grid, block_out); break;
            case 8: launch_eval<T, 8>(ctx, p, grid, block_out); break;
            case 16: launch_eval<T, 16>(ctx, p, grid, block_out); break;
            case 32: launch_eval<T, 32>(ctx, p, grid, block_out); break;
            default: launch_eval<T, 64>(ctx, p, grid, block_out); break;
        }
    }
    PMF_HIP_CHECK(hipGetLastError());
    PMF_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, block_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PMF_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double *h = (const double *)ctx->h_pinned;
    double s = 0.0;
    for (int l = 0; l < ev.n_labels; ++l) {
        abs_l[l] = 0.0;
        cnt_l[l] = 0;
    }
    for (int b = 0; b < grid; ++b) {
        const double *r = h + (size_t)b * rec;
        s += r[0];
        for (int l = 0; l < ev.n_labels; ++l) {
            abs_l[l] += r[1 + l];
            cnt_l[l] += (int64_t)r[1 + PMF_MAX_LABELS + l];
        }
    }
    *sse = s;
    return PMF_OK;
}

extern "C" int pmf_eval_run(pmf_ctx *ctx, int use_bias, double offset, double *sum_sq_err,
                            double *abs_err_per_label, int64_t *count_per_label) {
    PMF_REQUIRE(ctx != nullptr, PMF_EINVAL, "pmf_eval_run: null context");
    PMF_REQUIRE(sum_sq_err && abs_err_per_label && count_per_label, PMF_EINVAL, "pmf_eval_run: null argument");
    PMF_REQUIRE(ctx->eval.n > 0, PMF_EINVAL, "pmf_eval_run: no validation set (call pmf_eval_set)");
    PMF_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->dtype == PMF_F64) return run_eval<double>(ctx, use_bias, offset, sum_sq_err, abs_err_per_label, count_per_label);
    return run_eval<float>(ctx, use_bias, offset, sum_sq_err, abs_err_per_label, count_per_label);
}
